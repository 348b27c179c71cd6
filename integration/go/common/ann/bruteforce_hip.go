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
	"math"
	"sort"
	"sync"
	"unsafe"

	"github.com/gorse-io/gorse/common/encoding"
	"github.com/gorse-io/gorse/common/heap"
	"github.com/gorse-io/gorse/common/log"
	"github.com/pkg/errors"
	"github.com/samber/lo"
	"go.uber.org/zap"
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
	lastErr error // the device failure behind the last empty answer of SearchVector / SearchVectors, if any
}

// LastError returns (and clears) the device failure behind the last empty answer.
func (b *BruteforceHIP) LastError() error {
	b.mu.Lock()
	defer b.mu.Unlock()
	err := b.lastErr
	b.lastErr = nil
	return err
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

// SearchVector has no error result in ann.Index (ann.go:21-25): a device failure (no GPU, a dimension the library cannot hold)
// is logged and, like an empty index, answered with no neighbours -- LastError tells the two apart.
func (b *BruteforceHIP) SearchVector(q []float32, k int, prune0 bool) []lo.Tuple2[int, float32] {
	b.mu.Lock()
	defer b.mu.Unlock()
	if b.d == 0 || len(q) != b.d || k <= 0 {
		return nil
	}
	if err := b.sync(); err != nil {
		b.lastErr = err
		log.Logger().Error("BruteforceHIP: device index unavailable", zap.Error(err))
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

// The index section of a MatrixFactorizationItems file (logics/cf.go:81-128) in the place of HNSW.Marshal (hnsw.go:278-337).
// Marshal writes a section a build WITHOUT this library rejects cleanly: where HNSW.Unmarshal expects levelFactor it finds the
// bytes "GHIP", then an int64 version in the maxConnection slot, three int64 zeros, the vector count 1 and an EMPTY gob stream
// for that "vector" -- encoding.ReadGob returns gob's EOF there, an ordinary error, before anything is allocated.  Behind it:
// int64 count, int32 dimension, the float32 rows.  Unmarshal reads that section AND the reference's own (keeping the vectors,
// skipping the graph), so a master without the library can hand its file to a worker with it.
// The C++ twin of both directions, with its test on a hand-built reference stream: gorse_amd/host/gorse_vectors.hpp
// (logics::MatrixFactorizationItems), tests/test_items_blob_cpu.py.
var hipMagic = [4]byte{'G', 'H', 'I', 'P'}

func (b *BruteforceHIP) Marshal(w io.Writer) error {
	b.mu.Lock()
	defer b.mu.Unlock()
	n := 0
	if b.d > 0 {
		n = len(b.data) / b.d
	}
	head := struct {
		Magic   [4]byte
		Version int64
		Zero    [3]int64
		One     int64
		Empty   int32
		N       int64
		D       int32
	}{Magic: hipMagic, Version: 1, One: 1, N: int64(n), D: int32(b.d)}
	if err := binary.Write(w, binary.LittleEndian, head); err != nil {
		return errors.WithStack(err)
	}
	return errors.WithStack(binary.Write(w, binary.LittleEndian, b.data))
}

// MarshalReference writes the index section in the REFERENCE'S OWN format (HNSW.Marshal, hnsw.go:278-337) with a graph built by
// the device, for clusters in which some workers run a build without this library (set `recommend.collaborative.hip_compat_blob`
// on the master, see INTEGRATION.md): their HNSW.Unmarshal loads it and their own knnSearch walks it.  Every vector draws its
// level as insert does (hnsw.go:137); layer L holds the vectors of level >= L; a vector's queue in a layer = its nearest vectors
// OF THAT LAYER by the index's distance plus a few reverse links, 96 at the bottom and 48 above (NewHNSW, hnsw.go:52-60), from ONE exact all-pairs search
// per layer (gorse_topk_all_pairs) instead of one efConstruction search per insertion; queues ascending in the distance = valid
// heap arrays (heap/pq.go:42-48).  The C++ twin with its tests (recall 1.000 of the reference's search restated, 20,000 x 32):
// gorse_amd/host/gorse_vectors.hpp MarshalReference, tests/test_items_blob_cpu.py, tests/test_gpu_items_graph.py.
func (b *BruteforceHIP) MarshalReference(w io.Writer) error {
	b.mu.Lock()
	n := 0
	if b.d > 0 {
		n = len(b.data) / b.d
	}
	data, d := b.data, b.d
	b.mu.Unlock()
	const m, m0, efConstruction = 48, 96, 100
	levelFactor := float32(1.0 / math.Log(48))
	if err := binary.Write(w, binary.LittleEndian, struct {
		LevelFactor                float32
		M, M0, Ef, EfConstruction int64
		N                          int64
	}{levelFactor, m, m0, 0, efConstruction, int64(n)}); err != nil {
		return errors.WithStack(err)
	}
	for i := 0; i < n; i++ {
		if err := encoding.WriteGob(w, data[i*d:(i+1)*d]); err != nil {
			return errors.WithStack(err)
		}
	}
	// levels: ONE splitmix64 stream seeded by the count -- the very stream of the C++ twin (gorse_vectors.hpp MarshalReference), so that
	// both write the same blob for the same model, and the same model always yields the same blob (math/rand's global stream would not).
	// The arithmetic is the twin's, step for step (ReferenceLevels): log in float64, rounded to float32, ONE float32 multiplication by
	// levelFactor = float32(1 / math.Log(48)), floor in float64 -- a float32 logf on one side would differ by an ulp now and then, and a
	// product that straddles an integer would change a level and with it the blob (tests/test_items_blob_cpu.py pins the stream).
	level, top := make([]int, n), 0
	st := uint64(0x9E3779B97F4A7C15) ^ uint64(n)
	for i := range level {
		st += 0x9E3779B97F4A7C15
		z := st
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
		z = (z ^ (z >> 27)) * 0x94D049BB133111EB
		z ^= z >> 31
		u := (float32(z>>40) + 1) / 16777216 // (0, 1]
		level[i] = int(math.Floor(float64(-float32(math.Log(float64(u))) * levelFactor)))
		top = max(top, level[i])
	}
	// One layer, written as PriorityQueue.Marshal does.  A queue = the member's nearest limit - limit/8 other members (exact) + up to
	// limit/8 REVERSE links: members that list it among their nearest without being listed back, those with the fewest incoming links
	// first.  Under the inner product every exact list points at the same long vectors; a short vector nobody lists could never be
	// reached by knnSearch (insert links both ways, hnsw.go:163-183; its shrink keeps the nearest, which on exact lists would drop every
	// reverse link again).  Queues ascending in the distance.
	writeLayer := func(members []int32, limit int, withKeys bool) error {
		sub := NewBruteforceHIP(b.metric)
		defer sub.Close()
		for _, i := range members {
			sub.Add(data[int(i)*d : (int(i)+1)*d])
		}
		m := len(members)
		capR := 0
		if m > limit+1 {
			capR = limit / 8
		}
		capF := limit - capR
		k := min(capF+1, m) // + 1: the vector itself may be among its own nearest
		idx, dist, err := sub.SearchAll(k)
		if err != nil {
			return err
		}
		fwd := make([][]heap.Elem[int32, float32], m)
		indeg := make([]int32, m)
		for t := range members {
			for j := 0; j < k && len(fwd[t]) < capF; j++ {
				if r := idx[t*k+j]; r >= 0 && int(r) != t {
					fwd[t] = append(fwd[t], heap.Elem[int32, float32]{Value: r, Weight: dist[t*k+j]})
					indeg[r]++
				}
			}
		}
		rev := make([][]heap.Elem[int32, float32], m) // r <- t for every forward link t -> r
		if capR > 0 {
			for t := range members {
				for _, e := range fwd[t] {
					rev[e.Value] = append(rev[e.Value], heap.Elem[int32, float32]{Value: int32(t), Weight: e.Weight})
				}
			}
		}
		for r, key := range members {
			q := append([]heap.Elem[int32, float32]{}, fwd[r]...)
			if capR > 0 {
				cand := lo.Filter(rev[r], func(e heap.Elem[int32, float32], _ int) bool {
					return !lo.ContainsBy(fwd[r], func(f heap.Elem[int32, float32]) bool { return f.Value == e.Value })
				})
				sort.Slice(cand, func(a, b int) bool {
					if indeg[cand[a].Value] != indeg[cand[b].Value] {
						return indeg[cand[a].Value] < indeg[cand[b].Value]
					}
					if cand[a].Weight != cand[b].Weight {
						return cand[a].Weight < cand[b].Weight
					}
					return cand[a].Value < cand[b].Value
				})
				for _, e := range cand[:min(capR, len(cand))] {
					q = append(q, e)
					indeg[e.Value]++
				}
				sort.Slice(q, func(a, b int) bool {
					return q[a].Weight < q[b].Weight || (q[a].Weight == q[b].Weight && q[a].Value < q[b].Value)
				})
			}
			if withKeys {
				if err := binary.Write(w, binary.LittleEndian, key); err != nil {
					return errors.WithStack(err)
				}
			}
			elems := lo.Map(q, func(e heap.Elem[int32, float32], _ int) heap.Elem[int32, float32] {
				return heap.Elem[int32, float32]{Value: members[e.Value], Weight: e.Weight}
			})
			if err := binary.Write(w, binary.LittleEndian, false); err != nil {
				return errors.WithStack(err)
			}
			if err := encoding.WriteSlice(w, elems); err != nil {
				return errors.WithStack(err)
			}
		}
		return nil
	}
	all := make([]int32, n)
	for i := range all {
		all[i] = int32(i)
	}
	if n > 0 {
		if err := writeLayer(all, m0, false); err != nil {
			return err
		}
	}
	if err := binary.Write(w, binary.LittleEndian, int64(top)); err != nil {
		return errors.WithStack(err)
	}
	enter := int32(0)
	for l := 1; l <= top; l++ {
		members := lo.Filter(all, func(i int32, _ int) bool { return level[i] >= l })
		if err := binary.Write(w, binary.LittleEndian, int32(len(members))); err != nil {
			return errors.WithStack(err)
		}
		if err := writeLayer(members, m, true); err != nil {
			return err
		}
		if l == top {
			enter = members[0]
		}
	}
	return errors.WithStack(binary.Write(w, binary.LittleEndian, enter))
}

func skipQueue(r io.Reader) error { // PriorityQueue.Marshal, common/heap/pq.go:128-133: one bool, int32 length, 8-byte elements
	var head struct {
		Desc bool
		Len  int32
	}
	if err := binary.Read(r, binary.LittleEndian, &head); err != nil {
		return err
	}
	if head.Len < 0 {
		return errors.New("negative queue length")
	}
	_, err := io.CopyN(io.Discard, r, int64(head.Len)*8)
	return err
}

func (b *BruteforceHIP) Unmarshal(r io.Reader) error {
	b.mu.Lock()
	defer b.mu.Unlock()
	var first [4]byte
	if err := binary.Read(r, binary.LittleEndian, &first); err != nil {
		return errors.WithStack(err)
	}
	var params [4]int64
	if err := binary.Read(r, binary.LittleEndian, &params); err != nil {
		return errors.WithStack(err)
	}
	var n int64
	if err := binary.Read(r, binary.LittleEndian, &n); err != nil {
		return errors.WithStack(err)
	}
	if first == hipMagic { // this library's section
		if params[0] != 1 {
			return errors.Errorf("BruteforceHIP: index section version %d is not supported", params[0])
		}
		var rest struct {
			Empty int32
			N     int64
			D     int32
		}
		if err := binary.Read(r, binary.LittleEndian, &rest); err != nil {
			return errors.WithStack(err)
		}
		// One / Empty are what Marshal wrote (count 1, empty gob stream); N x D must be a size a sane file can hold: a corrupt
		// header must end in an error, not in a panic of make()
		if n != 1 || rest.Empty != 0 || rest.N < 0 || rest.D < 0 || rest.D > 1<<16 || rest.N > (1<<33)/int64(max(rest.D, 1)) {
			return errors.Errorf("BruteforceHIP: bad header (one %d, %+v)", n, rest)
		}
		b.d, b.data = int(rest.D), make([]float32, rest.N*int64(rest.D))
		b.dirty = true
		return errors.WithStack(binary.Read(r, binary.LittleEndian, b.data))
	}
	// the reference's section: the vectors, then a graph this index has no use for
	if n < 0 {
		return errors.New("negative vector count")
	}
	b.d, b.data = 0, b.data[:0]
	for i := int64(0); i < n; i++ {
		var v []float32
		if err := encoding.ReadGob(r, &v); err != nil {
			return errors.WithStack(err)
		}
		if b.d == 0 {
			b.d = len(v)
		} else if len(v) != b.d {
			return errors.Errorf("vector %d has %d dimensions, want %d", i, len(v), b.d)
		}
		b.data = append(b.data, v...)
	}
	for i := int64(0); i < n; i++ {
		if err := skipQueue(r); err != nil {
			return errors.WithStack(err)
		}
	}
	var layers int64
	if err := binary.Read(r, binary.LittleEndian, &layers); err != nil {
		return errors.WithStack(err)
	}
	for l := int64(0); l < layers; l++ {
		var m int32
		if err := binary.Read(r, binary.LittleEndian, &m); err != nil {
			return errors.WithStack(err)
		}
		for e := int32(0); e < m; e++ {
			var key int32
			if err := binary.Read(r, binary.LittleEndian, &key); err != nil {
				return errors.WithStack(err)
			}
			if err := skipQueue(r); err != nil {
				return errors.WithStack(err)
			}
		}
	}
	var enterPoint int32
	b.dirty = true
	return errors.WithStack(binary.Read(r, binary.LittleEndian, &enterPoint))
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

// SearchAllSharded is SearchAll over several GPUs of this process: one replica of the index per device, the TRIANGLE of the symmetric
// all-pairs sweep sharded over them (include/gorse_hip.h, gorse_topk_tri_*: device r takes the query blocks r, r + len(devices), ...; the
// pilot thresholds and the foreign candidate lists move between the devices inside the one library call).  Rows, distances and their
// order are SearchAll's, bit for bit (tests/test_gpu_topk_tri.py).  Measured per device, the devices emulated on one MI355X at a million
// 128-dimensional vectors: 122 / 71 / 46 ms at 2 / 4 / 8 devices against 222 ms on one; a search that has no symmetric form (fewer than
// 2^17 vectors, more than 128 bf16 / 42 fp32 dimensions) returns an error: fall back to SearchAll or to contiguous shards of SearchIndex.
func (b *BruteforceHIP) SearchAllSharded(k int, devices []int) (idx []int32, dist []float32, err error) {
	b.mu.Lock()
	defer b.mu.Unlock()
	n := 0
	if b.d > 0 {
		n = len(b.data) / b.d
	}
	if n == 0 || k <= 0 || len(devices) == 0 {
		return nil, nil, nil
	}
	hs := make([]*C.gorse_topk, len(devices))
	defer func() {
		for _, h := range hs {
			if h != nil {
				C.gorse_topk_destroy(h)
			}
		}
	}()
	for r, dev := range devices { // every device holds the whole index (a million x 128 fp32: 0.5 GB of its 288)
		if rc := C.gorse_topk_create(&hs[r], C.int32_t(dev), C.int64_t(n), C.int32_t(b.d), C.GORSE_DTYPE_F32, C.int32_t(b.metric),
			unsafe.Pointer(&b.data[0])); rc != 0 {
			return nil, nil, errors.Errorf("gorse_topk_create(device %d): %s", dev, C.GoString(C.gorse_hip_last_error()))
		}
	}
	idx, dist = make([]int32, n*k), make([]float32, n*k)
	if rc := C.gorse_topk_tri_all_pairs_local(&hs[0], C.int32_t(len(hs)), 0, C.int64_t(n), C.int32_t(k),
		(*C.int32_t)(unsafe.Pointer(&idx[0])), (*C.float)(unsafe.Pointer(&dist[0]))); rc != 0 {
		return nil, nil, errors.Errorf("gorse_topk_tri_all_pairs_local: %s", C.GoString(C.gorse_hip_last_error()))
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
