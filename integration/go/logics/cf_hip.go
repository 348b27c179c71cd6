//go:build cgo && hip

// logics/cf.go with the MI355X index.  The reference file changes in THREE lines (INTEGRATION.md, "logics/cf.go"): the field
// `index *ann.HNSW[[]float32]` becomes `index itemsIndex`, the constructor calls `newItemsIndex()`, and the non-hip build gets
// the one-function file that returns the HNSW (cf_nohip.go); the interface itself sits in the untagged cf_index.go.  Everything else -- Add, Search, Marshal, Unmarshal, the framing of the blob --
// stays the reference's code; only the index section of the blob differs, and BruteforceHIP.Unmarshal reads both forms
// (common/ann/bruteforce_hip.go).  Not compiled here: no Go toolchain in the build image.
package logics

import (
	"github.com/gorse-io/gorse/common/ann"
	"github.com/samber/lo"
)

// itemsIndex (the methods logics/cf.go calls on its index) is declared in cf_index.go, which carries no build tag: the
// non-hip build uses the type too.

type hipItemsIndex struct{ *ann.BruteforceHIP }

// Bruteforce.Add returns the NEW LENGTH (bruteforce.go:33-37), HNSW.Add the slot (hnsw.go:87-101)
func (x hipItemsIndex) Add(v []float32) int { return x.BruteforceHIP.Add(v) - 1 }

func newItemsIndex() itemsIndex { return hipItemsIndex{ann.NewBruteforceHIP(ann.NegDot)} }

// SearchBulk answers many users in ONE device search (gorse_topk_search_vector with nq queries: >= 768 of them run on the
// MFMA sweep): what the worker's per-user loop (worker/pipeline.go:403-448) becomes, see worker/pipeline_hip.go.
func (items *MatrixFactorizationItems) SearchBulk(vs [][]float32, n int) [][]lo.Tuple2[int, float32] {
	return items.index.(hipItemsIndex).SearchVectors(vs, n, false)
}
