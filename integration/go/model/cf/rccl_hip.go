//go:build cgo && hip

// Multi-GPU epoch bodies for the ONE goroutine that trains in the reference (master/tasks.go:879-1034): N gorse_mf handles,
// one per device, and N ranks of an RCCL communicator that libgorse_hip owns (gorse_comm_create_local).  The Go side holds no
// collective library: an exchange is ONE C call that receives all (handle, communicator) pairs and issues them as one RCCL
// group on the handles' own streams.
//
//	BPR  users sharded by contiguous row range (P rows and feedback lists local to their device, sampling local),
//	     Q replicated; per epoch every device runs its share of the samples, then gorse_mf_item_allreduce:
//	     Q <- Q_sync + sum over devices (Q - Q_sync)                     (I*d fp32; 102 MB at C3)
//	ALS  every device holds the dataset and both factor matrices and solves its row range of each half-sweep
//	     (gorse_als_set_ranges), then gorse_mf_rows_allgather moves the solved row blocks to every replica.
//
// Not compiled here (no Go toolchain in the build image); the Python twin of the same call sequence -- gorse_amd/dist.py
// LibComm + run_epoch / run_als_epoch -- is what tests/test_gpu_comm.py runs on the device.
package cf

/*
#include <stdlib.h>
#include "gorse_hip.h"
*/
import "C"

import (
	"unsafe"

	"github.com/gorse-io/gorse/dataset"
)

// shardRange is the contiguous row range [lo, hi) of rank r among n (remainder rows go to the first ranks): the split
// gorse_amd/dist.py shard_range uses, so that both hosts drive identical shards.
func shardRange(rows, r, n int) (lo, hi int) {
	base, rem := rows/n, rows%n
	lo = r*base + min(r, rem)
	hi = lo + base
	if r < rem {
		hi++
	}
	return
}

type hipGroup struct {
	handles []*C.gorse_mf
	comms   []*C.gorse_comm
	flatP   [][]float32 // per device: its user rows (BPR) or all rows (ALS), row-major
	flatQ   []float32
	uLo     []int // BPR: first user of every shard
	nUsers  int
	d       int
}

func (g *hipGroup) close() {
	for _, h := range g.handles {
		if h != nil {
			C.gorse_mf_destroy(h)
		}
	}
	for _, c := range g.comms {
		if c != nil {
			C.gorse_comm_destroy(c)
		}
	}
}

func (g *hipGroup) pairs() (**C.gorse_mf, **C.gorse_comm, C.int32_t) {
	return (**C.gorse_mf)(unsafe.Pointer(&g.handles[0])), (**C.gorse_comm)(unsafe.Pointer(&g.comms[0])), C.int32_t(len(g.handles))
}

// newHipGroupBPR shards the users of trainSet over the visible devices (or the first nDev of them).
func newHipGroupBPR(base *BaseMatrixFactorization, d int, trainSet dataset.CFSplit, nDev int) (*hipGroup, error) {
	var visible C.int32_t
	if rc := C.gorse_hip_device_count(&visible); rc != 0 {
		return nil, hipError("gorse_hip_device_count", rc)
	}
	if nDev <= 0 || nDev > int(visible) {
		nDev = int(visible)
	}
	g := &hipGroup{handles: make([]*C.gorse_mf, nDev), comms: make([]*C.gorse_comm, nDev), flatP: make([][]float32, nDev),
		uLo: make([]int, nDev+1), nUsers: trainSet.CountUsers(), d: d}
	devices := make([]C.int32_t, nDev)
	for r := range devices {
		devices[r] = C.int32_t(r)
	}
	if rc := C.gorse_comm_create_local(&g.comms[0], &devices[0], C.int32_t(nDev)); rc != 0 {
		return nil, hipError("gorse_comm_create_local", rc)
	}
	g.flatQ = packRows(base.ItemFactor, d)
	feedback := trainSet.GetUserFeedback()
	for r := 0; r < nDev; r++ {
		lo, hi := shardRange(g.nUsers, r, nDev)
		g.uLo[r], g.uLo[r+1] = lo, hi
		uptr, uidx := flatten(feedback[lo:hi])
		g.flatP[r] = packRows(base.UserFactor[lo:hi], d)
		if rc := C.gorse_mf_create(&g.handles[r], devices[r], C.int64_t(hi-lo), C.int64_t(trainSet.CountItems()), C.int32_t(d),
			(*C.int64_t)(unsafe.Pointer(&uptr[0])), (*C.int32_t)(unsafe.Pointer(&uidx[0])), nil, nil); rc != 0 {
			g.close()
			return nil, hipError("gorse_mf_create", rc)
		}
		if rc := C.gorse_mf_set_factors(g.handles[r], (*C.float)(unsafe.Pointer(&g.flatP[r][0])), (*C.float)(unsafe.Pointer(&g.flatQ[0]))); rc != 0 {
			g.close()
			return nil, hipError("gorse_mf_set_factors", rc)
		}
		if rc := C.gorse_mf_item_sync_mark(g.handles[r]); rc != 0 { // Q_sync <- Q: the base the deltas are taken against
			g.close()
			return nil, hipError("gorse_mf_item_sync_mark", rc)
		}
	}
	return g, nil
}

// bprEpoch is the epoch body of BPR.Fit (model.go:446-494) on all devices: every shard draws and applies its share of the
// CountFeedback() samples (the reference draws the user uniformly among users with feedback, model.go:452-458, so a shard's
// share is its share of such users), then the item factors are summed.  Nothing here waits for a device.
func (g *hipGroup) bprEpoch(trainSet dataset.CFSplit, lr, reg float32, seed uint64, epoch int) error {
	feedback := trainSet.GetUserFeedback()
	total := 0
	with := make([]int, len(g.handles))
	for r := range g.handles {
		for _, row := range feedback[g.uLo[r]:g.uLo[r+1]] {
			if len(row) > 0 {
				with[r]++
			}
		}
		total += with[r]
	}
	if total == 0 {
		return nil
	}
	for r, h := range g.handles {
		n := int64(float64(trainSet.CountFeedback())*float64(with[r])/float64(total) + 0.5)
		// sample_base r << 40: every shard reads its own stretch of the Philox stream
		if rc := C.gorse_bpr_epoch_enqueue(h, C.int64_t(n), C.float(lr), C.float(reg), C.uint64_t(seed), C.uint64_t(epoch),
			C.int64_t(r)<<40, C.GORSE_BPR_HOGWILD_STORES); rc != 0 {
			return hipError("gorse_bpr_epoch_enqueue", rc)
		}
	}
	hs, cs, n := g.pairs()
	if rc := C.gorse_mf_item_allreduce(hs, cs, n); rc != 0 {
		return hipError("gorse_mf_item_allreduce", rc)
	}
	return nil
}

// pullBPR copies every shard's user rows and (from device 0: the replicas are identical after an exchange) the item rows
// back into the model's [][]float32, whose rows alias flatP / flatQ.
func (g *hipGroup) pullBPR() {
	for r, h := range g.handles {
		var q *C.float
		if r == 0 {
			q = (*C.float)(unsafe.Pointer(&g.flatQ[0]))
		} else {
			scratch := make([]float32, len(g.flatQ))
			q = (*C.float)(unsafe.Pointer(&scratch[0]))
		}
		C.gorse_mf_get_factors(h, (*C.float)(unsafe.Pointer(&g.flatP[r][0])), q)
	}
}

// alsEpoch is the epoch body of ALS.Fit (model.go:641-738) with the rows of each half-sweep sharded: handles created over
// the WHOLE dataset (newHipModel on every device), ranges set once with gorse_als_set_ranges(shardRange(...)).
func alsEpochSharded(handles []*C.gorse_mf, comms []*C.gorse_comm, users, items int, weight, reg float32) error {
	n := len(handles)
	splits := func(rows int) []C.int64_t {
		s := make([]C.int64_t, n+1)
		for r := 0; r < n; r++ {
			lo, hi := shardRange(rows, r, n)
			s[r], s[r+1] = C.int64_t(lo), C.int64_t(hi)
		}
		return s
	}
	for side, rows := range []int{users, items} { // model.go:645-690, then :693-738
		// enqueued, not run: the synchronous gorse_als_half_epoch would make the N devices solve their row ranges one after
		// the other from this goroutine; the all-gather below is ordered behind the kernels on every handle's stream
		for _, h := range handles {
			if rc := C.gorse_als_half_epoch_enqueue(h, C.int32_t(side), C.float(weight), C.float(reg)); rc != 0 {
				return hipError("gorse_als_half_epoch_enqueue", rc)
			}
		}
		sp := splits(rows)
		if rc := C.gorse_mf_rows_allgather((**C.gorse_mf)(unsafe.Pointer(&handles[0])), (**C.gorse_comm)(unsafe.Pointer(&comms[0])),
			C.int32_t(n), C.int32_t(side), &sp[0]); rc != 0 {
			return hipError("gorse_mf_rows_allgather", rc)
		}
	}
	for _, h := range handles { // one host synchronisation per epoch
		if rc := C.gorse_mf_synchronize(h); rc != 0 {
			return hipError("gorse_mf_synchronize", rc)
		}
	}
	return nil
}
