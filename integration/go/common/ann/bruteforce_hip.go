//go:build cgo && hip

package ann

/*
#cgo LDFLAGS: -lgorse_hip
#include "gorse_hip.h"
*/
import "C"

import (
	"encoding/binary"
	"io"
	"sync"
	"unsafe"

	"github.com/pkg/errors"
	"github.com/samber/lo"
)

// Metric of a BruteforceHIP index (the distance functions the reference passes to ann.NewBruteforce / ann.NewHNSW).
type Metric int32

const (
	NegDot    Metric = C.GORSE_METRIC_NEG_DOT   // -floats.Dot            (logics/cf.go:32-34)
	Euclidean Metric = C.GORSE_METRIC_EUCLIDEAN // floats.Euclidean       (common/ann/ann_test.go)
	Cosine    Metric = C.GORSE_METRIC_COSINE    // 1 - a.b / (|a| |b|)
)

// BruteforceHIP is an exact ann.Index over []float32 resident on one MI355X.  Results equal Bruteforce[[]float32] with the
// matching distance function in every index and every bit, ties included (the library replays container/heap).
type BruteforceHIP struct {
	mu     sync.Mutex
	metric Metric
	d      int
	data   []float32 // row-major, len = n * d
	h      *C.gorse_topk
	dirty  bool
}

func NewBruteforceHIP(metric Metric) *BruteforceHIP { return &BruteforceHIP{metric: metric} }

// Add appends a vector and returns the new length (1-based like Bruteforce.Add, bruteforce.go:33-37).
func (b *BruteforceHIP) Add(v []float32) int {
	b.mu.Lock()
	defer b.mu.Unlock()
	if b.d == 0 {
		b.d = len(v)
	} else if len(v) != b.d {
		panic("floats: slice lengths do not match") // what the distance function would do on the first search
	}
	b.data = append(b.data, v...)
	b.dirty = true
	return len(b.data) / b.d
}

// sync (re)creates the device index after vectors were added.
func (b *BruteforceHIP) sync() error {
	if !b.dirty && b.h != nil {
		return nil
	}
	if b.h != nil {
		C.gorse_topk_destroy(b.h)
		b.h = nil
	}
	if rc := C.gorse_topk_create(&b.h, 0, C.int64_t(len(b.data)/b.d), C.int32_t(b.d), C.GORSE_DTYPE_F32, C.int32_t(b.metric),
		unsafe.Pointer(&b.data[0])); rc != 0 {
		return errors.Errorf("gorse_topk_create: %s", C.GoString(C.gorse_hip_last_error()))
	}
	b.dirty = false
	return nil
}

func zip(idx []int32, dist []float32, cnt int) []lo.Tuple2[int, float32] {
	out := make([]lo.Tuple2[int, float32], cnt)
	for t := 0; t < cnt; t++ {
		out[t] = lo.Tuple2[int, float32]{A: int(idx[t]), B: dist[t]}
	}
	return out
}

func cbool(v bool) C.int32_t {
	if v {
		return 1
	}
	return 0
}

func (b *BruteforceHIP) SearchIndex(q, k int, prune0 bool) ([]lo.Tuple2[int, float32], error) {
	b.mu.Lock()
	defer b.mu.Unlock()
	if b.d == 0 || q < 0 || q >= len(b.data)/b.d {
		return nil, errors.Errorf("index out of range: %v", q)
	}
	if k <= 0 { // the reference's heap of capacity 0 keeps nothing
		return []lo.Tuple2[int, float32]{}, nil
	}
	if err := b.sync(); err != nil {
		return nil, err
	}
	idx, dist := make([]int32, k), make([]float32, k)
	var cnt C.int32_t
	qq := C.int64_t(q)
	if rc := C.gorse_topk_search_index(b.h, &qq, 1, C.int32_t(k), cbool(prune0), (*C.int32_t)(unsafe.Pointer(&idx[0])),
		(*C.float)(unsafe.Pointer(&dist[0])), &cnt); rc != 0 {
		return nil, errors.Errorf("gorse_topk_search_index: %s", C.GoString(C.gorse_hip_last_error()))
	}
	return zip(idx, dist, int(cnt)), nil
}

func (b *BruteforceHIP) SearchVector(q []float32, k int, prune0 bool) []lo.Tuple2[int, float32] {
	b.mu.Lock()
	defer b.mu.Unlock()
	if b.d == 0 || len(q) != b.d || k <= 0 || b.sync() != nil {
		return nil
	}
	idx, dist := make([]int32, k), make([]float32, k)
	var cnt C.int32_t
	if rc := C.gorse_topk_search_vector(b.h, unsafe.Pointer(&q[0]), 1, C.int32_t(k), cbool(prune0), (*C.int32_t)(unsafe.Pointer(&idx[0])),
		(*C.float)(unsafe.Pointer(&dist[0])), &cnt); rc != 0 {
		return nil
	}
	return zip(idx, dist, int(cnt))
}

// SearchVectors is SearchVector for many queries in one device search (>= 768 queries take the MFMA sweep).  Queries whose
// length is not the index's dimension get an empty result, like SearchVector.
func (b *BruteforceHIP) SearchVectors(qs [][]float32, k int, prune0 bool) [][]lo.Tuple2[int, float32] {
	b.mu.Lock()
	defer b.mu.Unlock()
	out := make([][]lo.Tuple2[int, float32], len(qs))
	if b.d == 0 || k <= 0 || len(qs) == 0 || b.sync() != nil {
		return out
	}
	rows := make([]int, 0, len(qs))
	flat := make([]float32, 0, len(qs)*b.d)
	for t, q := range qs {
		if len(q) == b.d {
			rows = append(rows, t)
			flat = append(flat, q...)
		}
	}
	if len(rows) == 0 {
		return out
	}
	idx, dist, cnt := make([]int32, len(rows)*k), make([]float32, len(rows)*k), make([]C.int32_t, len(rows))
	if rc := C.gorse_topk_search_vector(b.h, unsafe.Pointer(&flat[0]), C.int64_t(len(rows)), C.int32_t(k), cbool(prune0),
		(*C.int32_t)(unsafe.Pointer(&idx[0])), (*C.float)(unsafe.Pointer(&dist[0])), &cnt[0]); rc != 0 {
		return out
	}
	for r, t := range rows {
		out[t] = zip(idx[r*k:(r+1)*k], dist[r*k:(r+1)*k], int(cnt[r]))
	}
	return out
}

// Marshal writes the vectors (dimension, count, row-major float32, little endian): the exact index has no graph to save.
func (b *BruteforceHIP) Marshal(w io.Writer) error {
	b.mu.Lock()
	defer b.mu.Unlock()
	n := 0
	if b.d > 0 {
		n = len(b.data) / b.d
	}
	if err := binary.Write(w, binary.LittleEndian, []int64{int64(b.metric), int64(b.d), int64(n)}); err != nil {
		return errors.WithStack(err)
	}
	return errors.WithStack(binary.Write(w, binary.LittleEndian, b.data))
}

// Unmarshal reads what Marshal wrote; the device index is rebuilt at the next search.
func (b *BruteforceHIP) Unmarshal(r io.Reader) error {
	b.mu.Lock()
	defer b.mu.Unlock()
	var head [3]int64
	if err := binary.Read(r, binary.LittleEndian, head[:]); err != nil {
		return errors.WithStack(err)
	}
	if head[1] < 0 || head[2] < 0 || (head[1] == 0 && head[2] != 0) {
		return errors.Errorf("BruteforceHIP: bad header %v", head)
	}
	b.metric, b.d = Metric(head[0]), int(head[1])
	b.data = make([]float32, head[1]*head[2])
	if err := binary.Read(r, binary.LittleEndian, b.data); err != nil {
		return errors.WithStack(err)
	}
	b.dirty = true
	return nil
}

// SearchAll is SearchIndex for every stored vector in one device pass (the item-to-item bulk build): row q of the
// results holds k (index, distance) pairs, padded with -1 / +Inf.
func (b *BruteforceHIP) SearchAll(k int) (idx []int32, dist []float32, err error) {
	b.mu.Lock()
	defer b.mu.Unlock()
	if b.d == 0 {
		return nil, nil, nil
	}
	if err = b.sync(); err != nil {
		return nil, nil, err
	}
	n := len(b.data) / b.d
	if n == 0 || k <= 0 {
		return nil, nil, nil
	}
	idx, dist = make([]int32, n*k), make([]float32, n*k)
	if rc := C.gorse_topk_all_pairs(b.h, 0, C.int64_t(n), C.int32_t(k), (*C.int32_t)(unsafe.Pointer(&idx[0])),
		(*C.float)(unsafe.Pointer(&dist[0]))); rc != 0 {
		return nil, nil, errors.Errorf("gorse_topk_all_pairs: %s", C.GoString(C.gorse_hip_last_error()))
	}
	return idx, dist, nil
}

// Close releases the device index.
func (b *BruteforceHIP) Close() {
	b.mu.Lock()
	defer b.mu.Unlock()
	if b.h != nil {
		C.gorse_topk_destroy(b.h)
		b.h = nil
	}
}
