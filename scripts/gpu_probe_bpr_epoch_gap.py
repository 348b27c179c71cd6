#!/usr/bin/env python3
"""Enqueued BPR epochs of the production schedule, back to back: wall time per epoch and the device time gorse_mf_epoch_times reports, per
shape and width -- what the packets between two epochs' kernels cost (round 6: an epoch that follows another begins at that one's end event,
and its end event is the last chunk's "consumed" event).  A/B against another build through scripts/gpu_ab_lib.sh.
usage: gpu_probe_bpr_epoch_gap.py [epochs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402

n_ep = int(sys.argv[1]) if len(sys.argv) > 1 else 48
sets = {"ml1m": synth.s_ml1m(), "ml100k": synth.s_ml100k()}
for shape, d in (("ml1m", 64), ("ml1m", 16), ("ml1m", 8), ("ml100k", 16), ("ml100k", 8)):
    data = sets[shape]
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 3)
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
    mf.set_factors(P0, Q0)
    mf.bpr_epoch(data.n_train, 0.05, 0.01, 7, 1, mode=capi.BPR_HOGWILD_STORES)
    for rep in range(3):
        mf.synchronize()
        mf.epoch_times(reset=True)
        t0 = time.perf_counter()
        for ep in range(1, n_ep + 1):
            mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 7, ep, mode=capi.BPR_HOGWILD_STORES)
        mf.synchronize()
        wall = (time.perf_counter() - t0) / n_ep * 1e3
        n, ms, fl = mf.epoch_times(reset=True)
        print("%-6s nFactors %3d: %.4f ms per epoch (wall, %d enqueued epochs) | device time of %d epochs %.4f ms each, %d in flight"
              % (shape, d, wall, n_ep, n, ms / max(n, 1), fl), flush=True)
    # the same through the Fit loop's pacing: at most two epochs in flight
    mf.synchronize()
    mf.epoch_times(reset=True)
    t0 = time.perf_counter()
    for ep in range(1, n_ep + 1):
        mf.epoch_throttle(2)
        mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 7, ep, mode=capi.BPR_HOGWILD_STORES)
    mf.synchronize()
    wall = (time.perf_counter() - t0) / n_ep * 1e3
    n, ms, fl = mf.epoch_times(reset=True)
    print("%-6s nFactors %3d: %.4f ms per epoch throttled to two in flight | device time of %d epochs %.4f ms each" % (shape, d, wall, n, ms / max(n, 1)), flush=True)
    del mf
