//go:build cgo && hip

package ann

/*
#cgo LDFLAGS: -lgorse_hip
#include "gorse_hip.h"
*/
import "C"

import (
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
	if b.d == 0 || len(q) != b.d || b.sync() != nil {
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
