//go:build cgo && hip

// logics/cf.go with the MI355X index: MatrixFactorizationItems keeps the same methods (Add, Search, Marshal, Unmarshal),
// but the neighbours come from ann.BruteforceHIP -- exact, on the device -- instead of ann.HNSW (logics/cf.go:32-62).  The
// non-hip build keeps the reference file (this one and logics/cf.go exclude each other through their build tags; the
// reference file gets `//go:build !cgo || !hip`).  Not compiled here: no Go toolchain in the build image.
package logics

import (
	"io"
	"sync"
	"time"

	"github.com/gorse-io/gorse/common/ann"
	"github.com/gorse-io/gorse/common/encoding"
	"github.com/gorse-io/gorse/common/log"
	"github.com/gorse-io/gorse/storage/cache"
	"github.com/pkg/errors"
	"github.com/samber/lo"
	"go.uber.org/zap"
)

type MatrixFactorizationItems struct {
	timestamp time.Time
	items     []string
	itemsLock sync.Mutex
	index     *ann.BruteforceHIP // distance = -floats.Dot (GORSE_METRIC_NEG_DOT), like `distance` of the reference file
	dimension int
}

func NewMatrixFactorizationItems(timestamp time.Time) *MatrixFactorizationItems {
	return &MatrixFactorizationItems{
		timestamp: timestamp,
		items:     make([]string, 0),
		index:     ann.NewBruteforceHIP(ann.NegDot),
	}
}

func (items *MatrixFactorizationItems) Add(itemId string, v []float32) {
	items.itemsLock.Lock()
	defer items.itemsLock.Unlock()
	if items.dimension == 0 {
		items.dimension = len(v)
	} else if items.dimension != len(v) {
		log.Logger().Error("dimension mismatch", zap.Int("dimension", len(v)))
		return
	}
	// Bruteforce.Add returns the NEW LENGTH (bruteforce.go:33-37; HNSW.Add returns the index): the id's slot is length - 1
	j := items.index.Add(v) - 1
	for len(items.items) <= j {
		items.items = append(items.items, "")
	}
	items.items[j] = itemId
}

func (items *MatrixFactorizationItems) Search(v []float32, n int) []cache.Score {
	scores := items.index.SearchVector(v, n, false)
	return lo.Map(scores, func(v lo.Tuple2[int, float32], _ int) cache.Score {
		return cache.Score{
			Id:        items.items[v.A],
			Score:     -float64(v.B),
			Timestamp: items.timestamp,
		}
	})
}

// SearchBulk answers many users in ONE device search (gorse_topk_search_vector with nq queries: >= 768 of them run on the
// MFMA sweep): what the worker's per-user loop (worker/pipeline.go:403-448) becomes, see worker/pipeline_hip.go.
func (items *MatrixFactorizationItems) SearchBulk(vs [][]float32, n int) [][]cache.Score {
	res := items.index.SearchVectors(vs, n, false)
	return lo.Map(res, func(scores []lo.Tuple2[int, float32], _ int) []cache.Score {
		return lo.Map(scores, func(v lo.Tuple2[int, float32], _ int) cache.Score {
			return cache.Score{Id: items.items[v.A], Score: -float64(v.B), Timestamp: items.timestamp}
		})
	})
}

// Marshal keeps the reference's framing (logics/cf.go:81-101): timestamp, dimension, the index, the ids.  The index part
// is the row-major vectors (BruteforceHIP.Marshal: count + float32 rows) instead of the HNSW graph: a file written by one
// build is read by the same build.
func (items *MatrixFactorizationItems) Marshal(w io.Writer) error {
	if err := encoding.WriteGob(w, items.timestamp); err != nil {
		return errors.WithStack(err)
	}
	if err := encoding.WriteGob(w, items.dimension); err != nil {
		return errors.WithStack(err)
	}
	if err := items.index.Marshal(w); err != nil {
		return errors.WithStack(err)
	}
	if err := encoding.WriteGob(w, int64(len(items.items))); err != nil {
		return errors.WithStack(err)
	}
	for _, item := range items.items {
		if err := encoding.WriteGob(w, item); err != nil {
			return errors.WithStack(err)
		}
	}
	return nil
}

func (items *MatrixFactorizationItems) Unmarshal(r io.Reader) error {
	if err := encoding.ReadGob(r, &items.timestamp); err != nil {
		return errors.WithStack(err)
	}
	if err := encoding.ReadGob(r, &items.dimension); err != nil {
		return errors.WithStack(err)
	}
	if err := items.index.Unmarshal(r); err != nil {
		return errors.WithStack(err)
	}
	var numItems int64
	if err := encoding.ReadGob(r, &numItems); err != nil {
		return errors.WithStack(err)
	}
	items.items = make([]string, numItems)
	for i := int64(0); i < numItems; i++ {
		if err := encoding.ReadGob(r, &items.items[i]); err != nil {
			return errors.WithStack(err)
		}
	}
	return nil
}
