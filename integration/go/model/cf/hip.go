//go:build cgo && hip

// Shared plumbing of the MI355X epoch bodies (bpr_hip.go, als_hip.go): the model's factors and the training set's
// feedback lists live on one GPU behind a gorse_mf handle for the duration of Fit.
package cf

/*
#cgo LDFLAGS: -lgorse_hip
#include <stdlib.h>
#include "gorse_hip.h"
*/
import "C"

import (
	"context"
	"math"
	"sync/atomic"
	"unsafe"

	mapset "github.com/deckarep/golang-set/v2"
	"github.com/gorse-io/gorse/common/floats"
	"github.com/gorse-io/gorse/dataset"
	"github.com/pkg/errors"
	"github.com/samber/lo"
)

func hipError(what string, rc C.int32_t) error {
	return errors.Errorf("%s: %s (code %d)", what, C.GoString(C.gorse_hip_last_error()), int(rc))
}

// flatten turns [][]int32 into CSR; the order inside a row is kept (the positive pick UserFeedback[u][Intn(n)] reads it).
func flatten(rows [][]int32) (indptr []int64, indices []int32) {
	indptr = make([]int64, len(rows)+1)
	for i, r := range rows {
		indptr[i+1] = indptr[i] + int64(len(r))
	}
	indices = make([]int32, 0, indptr[len(rows)]+1)
	for _, r := range rows {
		indices = append(indices, r...)
	}
	if len(indices) == 0 {
		indices = append(indices, 0) // &indices[0] must exist
	}
	return
}

// packRows re-homes the rows of a [][]float32 in ONE backing array and makes every row a slice of it, so that
// gorse_mf_get_factors fills the model's rows in place.
func packRows(rows [][]float32, d int) []float32 {
	flat := make([]float32, len(rows)*d+1)
	for i := range rows {
		copy(flat[i*d:(i+1)*d], rows[i])
		rows[i] = flat[i*d : (i+1)*d : (i+1)*d]
	}
	return flat
}

type hipModel struct {
	h            *C.gorse_mf
	flatP, flatQ []float32
	negativesResident bool // gorse_mf_sample_user_negatives has run on this handle (evaluateResident)
	cancel       *C.int32_t // C memory: a goroutine sets it when ctx is done, the library polls it between launches
	stop         chan struct{}
}

func newHipModel(ctx context.Context, base *BaseMatrixFactorization, d int, trainSet dataset.CFSplit, withItems bool) (*hipModel, error) {
	m := &hipModel{stop: make(chan struct{})}
	m.flatP, m.flatQ = packRows(base.UserFactor, d), packRows(base.ItemFactor, d)
	uptr, uidx := flatten(trainSet.GetUserFeedback())
	var iptrP *C.int64_t
	var iidxP *C.int32_t
	if withItems {
		iptr, iidx := flatten(trainSet.GetItemFeedback())
		iptrP, iidxP = (*C.int64_t)(unsafe.Pointer(&iptr[0])), (*C.int32_t)(unsafe.Pointer(&iidx[0]))
	}
	if rc := C.gorse_mf_create(&m.h, 0, C.int64_t(trainSet.CountUsers()), C.int64_t(trainSet.CountItems()), C.int32_t(d),
		(*C.int64_t)(unsafe.Pointer(&uptr[0])), (*C.int32_t)(unsafe.Pointer(&uidx[0])), iptrP, iidxP); rc != 0 {
		return nil, hipError("gorse_mf_create", rc)
	}
	if rc := C.gorse_mf_set_factors(m.h, (*C.float)(unsafe.Pointer(&m.flatP[0])), (*C.float)(unsafe.Pointer(&m.flatQ[0]))); rc != 0 {
		C.gorse_mf_destroy(m.h)
		return nil, hipError("gorse_mf_set_factors", rc)
	}
	m.cancel = (*C.int32_t)(C.calloc(1, 4))
	go func() {
		select {
		case <-ctx.Done():
			atomic.StoreInt32((*int32)(unsafe.Pointer(m.cancel)), 1)
		case <-m.stop:
		}
	}()
	return m, nil
}

// pull copies the factors back into the model's rows (they alias flatP / flatQ).
func (m *hipModel) pull() {
	C.gorse_mf_get_factors(m.h, (*C.float)(unsafe.Pointer(&m.flatP[0])), (*C.float)(unsafe.Pointer(&m.flatQ[0])))
}

func (m *hipModel) close() {
	close(m.stop)
	C.gorse_mf_destroy(m.h)
	C.free(unsafe.Pointer(m.cancel))
}

// evaluate is Evaluate (evaluator.go:35-72) with Rank replaced by ONE gorse_mf_rank call over all test users; negatives
// and metric functions stay the reference's.
func (m *hipModel) evaluate(testSet, trainSet dataset.CFSplit, topK, numCandidates int, scorers ...Metric) []float32 {
	negatives := testSet.SampleUserNegatives(trainSet, numCandidates)
	users := make([]int32, 0, testSet.CountUsers())
	candPtr := []int64{0}
	cand := make([]int32, 0)
	for u, positives := range testSet.GetUserFeedback() {
		if len(positives) == 0 {
			continue
		}
		users = append(users, int32(u))
		cand = append(cand, positives...)
		cand = append(cand, negatives[u]...)
		candPtr = append(candPtr, int64(len(cand)))
	}
	sum := make([]float32, len(scorers))
	if len(users) == 0 {
		return sum
	}
	rank := make([]int32, len(users)*topK)
	rankLen := make([]int32, len(users))
	if rc := C.gorse_mf_rank(m.h, C.int64_t(len(users)), (*C.int32_t)(unsafe.Pointer(&users[0])), (*C.int64_t)(unsafe.Pointer(&candPtr[0])),
		(*C.int32_t)(unsafe.Pointer(&cand[0])), C.int32_t(topK), (*C.int32_t)(unsafe.Pointer(&rank[0])), (*C.int32_t)(unsafe.Pointer(&rankLen[0]))); rc != 0 {
		panic(hipError("gorse_mf_rank", rc))
	}
	for t, u := range users {
		targetSet := mapset.NewSet(testSet.GetUserFeedback()[u]...)
		rankList := rank[t*topK : t*topK+int(rankLen[t])]
		for i, metric := range scorers {
			sum[i] += metric(targetSet, rankList)
		}
	}
	floats.MulConst(sum, 1/float32(len(users)))
	return sum
}

// evaluateResident is the same Evaluate with the negatives drawn ON THE DEVICE (gorse_mf_sample_user_negatives: dataset.go:242-253
// over RandomGenerator.SampleInt32, one Philox stream per test user -- the host's RandomGenerator(seed, epoch, sample) in the C++
// twin draws the same lists) and the candidate lists left there: the first call of a Fit samples, every later evaluation only ranks
// (gorse_mf_rank_resident), so nothing but topK ids per user crosses PCIe between two epochs.  The reference caches the negatives in
// the Dataset the same way (dataset.go:243: drawn once, `if len(d.negatives) == 0`).
func (m *hipModel) evaluateResident(testSet, trainSet dataset.CFSplit, topK, numCandidates int, seed uint64, scorers ...Metric) []float32 {
	if !m.negativesResident {
		// (flatten leaves one unused word in `indices` when the split holds no feedback: &indices[0] exists)
		indptr, indices := flatten(testSet.GetUserFeedback())
		if rc := C.gorse_mf_sample_user_negatives(m.h, (*C.int64_t)(unsafe.Pointer(&indptr[0])), (*C.int32_t)(unsafe.Pointer(&indices[0])),
			C.int32_t(numCandidates), C.uint64_t(seed), nil, nil); rc != 0 {
			panic(hipError("gorse_mf_sample_user_negatives", rc))
		}
		m.negativesResident = true
	}
	var nUsers, nCand C.int64_t
	if rc := C.gorse_mf_resident_candidates(m.h, &nUsers, &nCand); rc != 0 {
		panic(hipError("gorse_mf_resident_candidates", rc))
	}
	sum := make([]float32, len(scorers))
	if nUsers == 0 {
		// a validation split without feedback: the reference's Evaluate averages over zero users -- floats.MulConst(sum, 1/count)
		// with count = 0 is 0 * +Inf = NaN (evaluator.go:69-70) -- and so does this twin: early stopping and the log lines of Fit
		// then see what they see in the reference
		for i := range sum {
			sum[i] = float32(math.NaN())
		}
		return sum
	}
	users := make([]int32, int(nUsers))
	rank := make([]int32, int(nUsers)*topK)
	rankLen := make([]int32, int(nUsers))
	if rc := C.gorse_mf_rank_resident(m.h, C.int32_t(topK), (*C.int32_t)(unsafe.Pointer(&users[0])), (*C.int32_t)(unsafe.Pointer(&rank[0])),
		(*C.int32_t)(unsafe.Pointer(&rankLen[0]))); rc != 0 {
		panic(hipError("gorse_mf_rank_resident", rc))
	}
	for t, u := range users {
		targetSet := mapset.NewSet(testSet.GetUserFeedback()[u]...)
		rankList := rank[t*topK : t*topK+int(rankLen[t])]
		for i, metric := range scorers {
			sum[i] += metric(targetSet, rankList)
		}
	}
	floats.MulConst(sum, 1/float32(len(users)))
	return sum
}

// earlyStop is the patience rule of model.go:508-517 on the recorded (epoch, NDCG) pairs.
func earlyStop(scores []lo.Tuple2[int, float32], epoch, patience int) (lo.Tuple2[int, float32], bool) {
	if patience <= 0 || epoch <= patience {
		return lo.Tuple2[int, float32]{}, false
	}
	best := lo.MaxBy(scores, func(a, b lo.Tuple2[int, float32]) bool { return a.B > b.B })
	return best, best.A <= epoch-patience
}
