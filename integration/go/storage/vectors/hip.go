//go:build cgo && hip

package vectors

/*
#cgo LDFLAGS: -lgorse_hip
#include "gorse_hip.h"
*/
import "C"

import (
	"context"
	"fmt"
	"slices"
	"sort"
	"sync"
	"time"
	"unsafe"

	"github.com/gorse-io/gorse/storage"
	"github.com/pkg/errors"
)

// HipPrefix selects this backend: vector database URLs "hip://" keep every collection in host memory and search it
// EXACTLY on one MI355X (dense: gorse_topk_*, sparse: gorse_sparse_*).  The semantics are the xvec backend's
// (xvec.go): upsert by Id, millisecond cut-off deletes, hidden / CONTAIN_ALL filters, Score = the inner product for Dot
// and the negated distance otherwise, zero scores dropped from sparse results after the cut to topK.
const HipPrefix = "hip://"

func init() {
	Register([]string{HipPrefix}, func(path, tablePrefix string, _ ...storage.Option) (Database, error) {
		return &Hip{collections: make(map[string]*hipCollection)}, nil
	})
}

type hipCollection struct {
	info CollectionInfo
	rows []Vector
	byID map[string]int
	// dense: rows x Dimension, row-major, and its device index (rebuilt lazily after a change)
	data  []float32
	dense *C.gorse_topk
	// sparse: the rows as CSR with ascending indices, and its device index
	indptr  []int64
	indices []uint32
	values  []float32
	sparse  *C.gorse_sparse
}

func (c *hipCollection) dropIndex() {
	if c.dense != nil {
		C.gorse_topk_destroy(c.dense)
		c.dense = nil
	}
	if c.sparse != nil {
		C.gorse_sparse_destroy(c.sparse)
		c.sparse = nil
	}
	c.indptr = nil
}

// Hip implements Database.  One mutex serialises everything: a device handle takes one caller at a time.
type Hip struct {
	mu          sync.Mutex
	collections map[string]*hipCollection
	closed      bool
}

func (db *Hip) Init() error { return nil }

func (db *Hip) Optimize(_ context.Context, _ string) error { return nil }

func lastError(what string) error {
	return errors.Errorf("%s: %s", what, C.GoString(C.gorse_hip_last_error()))
}

// admissible: hidden vectors never match; `categories` is CONTAIN_ALL (xvec.go:386-394)
func admissible(v *Vector, categories []string) bool {
	if v.IsHidden {
		return false
	}
	for _, c := range categories {
		if !slices.Contains(v.Categories, c) {
			return false
		}
	}
	return true
}

func (db *Hip) Close() error {
	db.mu.Lock()
	defer db.mu.Unlock()
	for _, c := range db.collections {
		c.dropIndex()
	}
	db.closed = true
	return nil
}

func (db *Hip) coll(ctx context.Context, name string) (*hipCollection, error) {
	if err := ctx.Err(); err != nil {
		return nil, errors.WithStack(err)
	}
	if db.closed {
		return nil, errors.New("hip vector database is closed")
	}
	c, ok := db.collections[name]
	if !ok {
		return nil, errors.Wrapf(storage.ErrNotFound, "collection %s", name)
	}
	return c, nil
}

func (db *Hip) ListCollections(ctx context.Context) ([]string, error) {
	db.mu.Lock()
	defer db.mu.Unlock()
	names := make([]string, 0, len(db.collections))
	for name := range db.collections {
		names = append(names, name)
	}
	sort.Strings(names)
	return names, ctx.Err()
}

func (db *Hip) DescribeCollection(ctx context.Context, name string) (*CollectionInfo, error) {
	db.mu.Lock()
	defer db.mu.Unlock()
	c, err := db.coll(ctx, name)
	if err != nil {
		return nil, err
	}
	info := c.info
	return &info, nil
}

func (db *Hip) AddCollection(ctx context.Context, name string, dimensions int, distance Distance, config VectorConfig) error {
	db.mu.Lock()
	defer db.mu.Unlock()
	if err := ctx.Err(); err != nil {
		return errors.WithStack(err)
	}
	if dimensions < 0 {
		return errors.Errorf("invalid vector dimension %d", dimensions)
	}
	if config.Type != QuantizationNone {
		return fmt.Errorf("quantization type %s for hip %w", config.Type, storage.ErrNotSupported)
	}
	if dimensions == 0 && distance != Dot { // xvec.go:243-245
		return fmt.Errorf("distance method for sparse vector %w", storage.ErrNotSupported)
	}
	if _, exists := db.collections[name]; exists {
		return errors.Wrapf(storage.ErrAlreadyExists, "collection %s", name)
	}
	db.collections[name] = &hipCollection{
		info: CollectionInfo{Name: name, Dimension: dimensions, Distance: distance, VectorConfig: config},
		byID: make(map[string]int),
	}
	return nil
}

func (db *Hip) DeleteCollection(ctx context.Context, name string) error {
	db.mu.Lock()
	defer db.mu.Unlock()
	c, err := db.coll(ctx, name)
	if err != nil {
		return err
	}
	c.dropIndex()
	delete(db.collections, name)
	return nil
}

func (db *Hip) CountVectors(ctx context.Context, name string) (int64, error) {
	db.mu.Lock()
	defer db.mu.Unlock()
	c, err := db.coll(ctx, name)
	if err != nil {
		return 0, err
	}
	return int64(len(c.rows)), nil
}

func (db *Hip) AddVectors(ctx context.Context, name string, vectors []Vector) error {
	if len(vectors) == 0 {
		return nil
	}
	db.mu.Lock()
	defer db.mu.Unlock()
	c, err := db.coll(ctx, name)
	if err != nil {
		return err
	}
	sparse := c.info.Dimension == 0
	for _, v := range vectors { // validate everything before touching the collection (xvec.go:318-321)
		if sparse {
			if len(v.Indices) == 0 || len(v.Indices) != len(v.Values) {
				return errors.Errorf("vector %s is not a sparse vector", v.Id)
			}
			sorted := slices.Clone(v.Indices)
			slices.Sort(sorted)
			if len(slices.Compact(sorted)) != len(v.Indices) {
				return errors.Errorf("vector %s repeats an index", v.Id)
			}
		} else if len(v.Indices) != 0 || len(v.Values) != c.info.Dimension {
			return errors.Errorf("vector %s has dimension %d, collection %s has %d", v.Id, len(v.Values), name, c.info.Dimension)
		}
	}
	for _, v := range vectors {
		v.Timestamp = v.Timestamp.Truncate(time.Millisecond) // what the file-backed backends keep
		if at, exists := c.byID[v.Id]; exists {             // upsert
			c.rows[at] = v
			if !sparse {
				copy(c.data[at*c.info.Dimension:], v.Values)
			}
		} else {
			c.byID[v.Id] = len(c.rows)
			c.rows = append(c.rows, v)
			if !sparse {
				c.data = append(c.data, v.Values...)
			}
		}
	}
	c.dropIndex()
	return nil
}

func (db *Hip) GetVectors(ctx context.Context, name string, ids []string) ([]Vector, error) {
	db.mu.Lock()
	defer db.mu.Unlock()
	c, err := db.coll(ctx, name)
	if err != nil {
		return nil, err
	}
	found := make([]Vector, 0, len(ids))
	for _, id := range ids {
		if at, exists := c.byID[id]; exists {
			found = append(found, c.rows[at])
		}
	}
	return orderVectors(ids, found), nil
}

func (db *Hip) DeleteVectors(ctx context.Context, name string, timestamp time.Time) error {
	db.mu.Lock()
	defer db.mu.Unlock()
	c, err := db.coll(ctx, name)
	if err != nil {
		return err
	}
	cutoff := timestamp.UnixMilli()
	keep := c.rows[:0]
	for _, v := range c.rows {
		if v.Timestamp.UnixMilli() >= cutoff { // xvec.go:371-377: timestamp < cutoff is deleted
			keep = append(keep, v)
		}
	}
	if len(keep) == len(c.rows) {
		return nil
	}
	c.rows = keep
	c.byID = make(map[string]int, len(keep))
	c.data = c.data[:0]
	for at, v := range c.rows {
		c.byID[v.Id] = at
		if c.info.Dimension != 0 {
			c.data = append(c.data, v.Values...)
		}
	}
	c.dropIndex()
	return nil
}

func (db *Hip) QueryVectors(ctx context.Context, name string, q Vector, categories []string, topK int) ([]ScoredVector, error) {
	if topK <= 0 {
		return []ScoredVector{}, nil
	}
	db.mu.Lock()
	defer db.mu.Unlock()
	c, err := db.coll(ctx, name)
	if err != nil {
		return nil, err
	}
	if len(c.rows) == 0 {
		return []ScoredVector{}, nil
	}
	if len(q.Indices) > 0 {
		res, err := db.querySparse(c, []Vector{q}, categories, topK)
		if err != nil {
			return nil, err
		}
		return res[0], nil
	}
	return db.queryDense(c, q.Values, categories, topK)
}

// queryDense: the exact top-K of the admissible vectors in ONE search: the filter (hidden vectors, CONTAIN_ALL categories,
// xvec.go:386-394) goes to the device as a mask (gorse_topk_set_mask), so k stays topK however selective the filter is.
func (db *Hip) queryDense(c *hipCollection, q []float32, categories []string, topK int) ([]ScoredVector, error) {
	d, n := c.info.Dimension, len(c.rows)
	if d == 0 || len(q) != d {
		return nil, errors.Errorf("query has dimension %d, collection %s has %d", len(q), c.info.Name, d)
	}
	if n == 0 || topK <= 0 {
		return []ScoredVector{}, nil
	}
	if c.dense == nil {
		metric := C.int32_t(C.GORSE_METRIC_COSINE)
		switch c.info.Distance {
		case Dot:
			metric = C.GORSE_METRIC_NEG_DOT
		case Euclidean:
			metric = C.GORSE_METRIC_EUCLIDEAN
		}
		if rc := C.gorse_topk_create(&c.dense, 0, C.int64_t(n), C.int32_t(d), C.GORSE_DTYPE_F32, metric, unsafe.Pointer(&c.data[0])); rc != 0 {
			return nil, lastError("gorse_topk_create")
		}
	}
	ok := make([]uint8, n)
	for t := range c.rows {
		if admissible(&c.rows[t], categories) {
			ok[t] = 1
		}
	}
	if rc := C.gorse_topk_set_mask(c.dense, (*C.uint8_t)(unsafe.Pointer(&ok[0]))); rc != 0 {
		return nil, lastError("gorse_topk_set_mask")
	}
	k := min(n, topK)
	idx, dist := make([]int32, k), make([]float32, k)
	var cnt C.int32_t
	if rc := C.gorse_topk_search_vector(c.dense, unsafe.Pointer(&q[0]), 1, C.int32_t(k), 0, (*C.int32_t)(unsafe.Pointer(&idx[0])),
		(*C.float)(unsafe.Pointer(&dist[0])), &cnt); rc != 0 {
		return nil, lastError("gorse_topk_search_vector")
	}
	results := make([]ScoredVector, 0, int(cnt))
	for t := 0; t < int(cnt); t++ {
		results = append(results, ScoredVector{Vector: c.rows[idx[t]], Score: -dist[t]}) // Dot: a.b; else the negated distance
	}
	return results, nil
}

// querySparse answers all queries in ONE device call; the filter travels as an admissibility mask.
func (db *Hip) querySparse(c *hipCollection, queries []Vector, categories []string, topK int) ([][]ScoredVector, error) {
	if c.info.Dimension != 0 {
		return nil, errors.Errorf("sparse query against the dense collection %s", c.info.Name)
	}
	if topK > 1024 {
		return nil, fmt.Errorf("topK > 1024 on a sparse collection for hip %w", storage.ErrNotSupported)
	}
	appendSorted := func(v *Vector, indices []uint32, values []float32) ([]uint32, []float32) {
		order := make([]int, len(v.Indices))
		for t := range order {
			order[t] = t
		}
		sort.Slice(order, func(a, b int) bool { return v.Indices[order[a]] < v.Indices[order[b]] })
		for _, t := range order {
			indices, values = append(indices, v.Indices[t]), append(values, v.Values[t])
		}
		return indices, values
	}
	if c.indptr == nil {
		c.indptr, c.indices, c.values = []int64{0}, c.indices[:0], c.values[:0]
		for t := range c.rows {
			c.indices, c.values = appendSorted(&c.rows[t], c.indices, c.values)
			c.indptr = append(c.indptr, int64(len(c.indices)))
		}
	}
	if c.sparse == nil {
		if rc := C.gorse_sparse_create(&c.sparse, 0, C.int64_t(len(c.rows)), (*C.int64_t)(unsafe.Pointer(&c.indptr[0])),
			(*C.uint32_t)(unsafe.Pointer(&c.indices[0])), (*C.float)(unsafe.Pointer(&c.values[0]))); rc != 0 {
			return nil, lastError("gorse_sparse_create")
		}
	}
	ok := make([]uint8, len(c.rows))
	for t := range c.rows {
		if admissible(&c.rows[t], categories) {
			ok[t] = 1
		}
	}
	if rc := C.gorse_sparse_set_mask(c.sparse, (*C.uint8_t)(unsafe.Pointer(&ok[0]))); rc != 0 {
		return nil, lastError("gorse_sparse_set_mask")
	}
	qptr, qidx, qval := []int64{0}, make([]uint32, 0, 64), make([]float32, 0, 64)
	for t := range queries {
		if len(queries[t].Indices) != len(queries[t].Values) {
			return nil, errors.New("sparse query: Indices and Values differ in length")
		}
		qidx, qval = appendSorted(&queries[t], qidx, qval)
		qptr = append(qptr, int64(len(qidx)))
	}
	if len(qidx) == 0 { // &qidx[0] must exist
		qidx, qval = append(qidx, 0), append(qval, 0)
	}
	nq := len(queries)
	if nq == 0 || topK <= 0 {
		return make([][]ScoredVector, nq), nil
	}
	idx, score, cnt := make([]int32, nq*topK), make([]float32, nq*topK), make([]int32, nq)
	if rc := C.gorse_sparse_search(c.sparse, C.int64_t(nq), (*C.int64_t)(unsafe.Pointer(&qptr[0])), (*C.uint32_t)(unsafe.Pointer(&qidx[0])),
		(*C.float)(unsafe.Pointer(&qval[0])), nil, C.int32_t(topK), (*C.int32_t)(unsafe.Pointer(&idx[0])),
		(*C.float)(unsafe.Pointer(&score[0])), (*C.int32_t)(unsafe.Pointer(&cnt[0]))); rc != 0 {
		return nil, lastError("gorse_sparse_search")
	}
	out := make([][]ScoredVector, nq)
	for t := 0; t < nq; t++ {
		out[t] = make([]ScoredVector, 0, cnt[t])
		for e := 0; e < int(cnt[t]); e++ {
			out[t] = append(out[t], ScoredVector{Vector: c.rows[idx[t*topK+e]], Score: score[t*topK+e]})
		}
	}
	return out, nil
}

// QuerySparseBatch is the bulk form the similarity refresh uses (every item / user of a sparse kind in one device search).
func (db *Hip) QuerySparseBatch(ctx context.Context, name string, queries []Vector, categories []string, topK int) ([][]ScoredVector, error) {
	db.mu.Lock()
	defer db.mu.Unlock()
	c, err := db.coll(ctx, name)
	if err != nil {
		return nil, err
	}
	if topK <= 0 || len(queries) == 0 || len(c.rows) == 0 {
		return make([][]ScoredVector, len(queries)), nil
	}
	return db.querySparse(c, queries, categories, topK)
}
