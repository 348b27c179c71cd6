#!/usr/bin/env python3
"""The user-run schedule with a user's run cut into segments (csrc/bpr.hip, SEG; nFactors <= 32), at the reference's own training
shape (S-ml1m, 30 epochs, lr 0.05, reg 0.01, init N(0, 0.001): model/cf/model_test.go:35-48): ms per epoch and NDCG@10 over three
sampler seeds per setting, next to the sequential oracle's NDCG.  usage: gpu_probe_bpr_segments.py [widths ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

widths = [int(a) for a in sys.argv[1:]] or [8, 16, 32]
data = synth.s_ml1m()
o = orc.Oracle()
epochs, lr, reg = 30, 0.05, 0.01
L = capi.lib()
srt = orc.sort_rows(data.uptr, data.uidx)
for d in widths:
    P, Q = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 3)
    t0 = time.perf_counter()
    for ep in range(1, epochs + 1):
        o.bpr_epoch_sampled(P, Q, data.uptr, data.uidx, srt, 77, ep, 0, data.n_train, lr, reg)
    ref = float(o.evaluate(P, Q, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0])
    print("nFactors %d: sequential oracle NDCG@10 %.4f (%.1f s)" % (d, ref, time.perf_counter() - t0), flush=True)
    for segs in (1, 2, 3, 4, 6, 8):
        L.gorse_hip_test_set_bpr_user_segments(segs)
        nd, ms = [], []
        for seed in (11, 22, 33):
            P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, seed)
            mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
            mf.set_factors(P0, Q0)
            mf.bpr_epoch(data.n_train, lr, reg, seed, 1, mode=capi.BPR_HOGWILD_STORES)  # code objects, buffers
            mf.set_factors(P0, Q0)
            t0 = time.perf_counter()
            for ep in range(1, epochs + 1):
                mf.bpr_epoch_enqueue(data.n_train, lr, reg, seed, ep, mode=capi.BPR_HOGWILD_STORES)
            mf.synchronize()
            ms.append((time.perf_counter() - t0) / epochs * 1e3)
            gP, gQ = mf.get_factors()
            nd.append(float(o.evaluate(gP, gQ, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0]))
            mf.close()
        print("  %d segment(s) per user: %.3f ms per epoch (min %.3f)  NDCG@10 %s  mean %.4f (oracle %+.4f)"
              % (segs, float(np.mean(ms)), min(ms), " ".join("%.4f" % x for x in nd), float(np.mean(nd)), float(np.mean(nd)) - ref), flush=True)
    L.gorse_hip_test_set_bpr_user_segments(0)
