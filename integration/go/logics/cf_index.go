// No build tag: both the hip and the non-hip build of package logics use this type (logics/cf.go's field `index itemsIndex`,
// cf_hip.go's and cf_nohip.go's newItemsIndex).  Not compiled here: no Go toolchain in the build image.
package logics

import (
	"io"

	"github.com/samber/lo"
)

// itemsIndex is what MatrixFactorizationItems needs of its index (the methods logics/cf.go:36-128 calls on ann.HNSW).
type itemsIndex interface {
	Add(v []float32) int // the slot of the new vector
	SearchVector(q []float32, n int, prune0 bool) []lo.Tuple2[int, float32]
	Marshal(w io.Writer) error
	Unmarshal(r io.Reader) error
}
