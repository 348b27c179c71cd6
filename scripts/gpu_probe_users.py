#!/usr/bin/env python3
"""GPU probe: the two Hogwild schedules of the BPR update -- per-sample groups (bpr_update_kernel) vs user runs
(bpr_update_user_kernel: counting sort by user, p_u register-resident over a user's samples) -- on the S-ml1m,
S-ml100k and C3-shard shapes: update / sort / sampler time per epoch, end-to-end epoch rate, and NDCG@10 after 8
epochs next to the sequential CPU oracle.  Output -> profiles/rNN_*_probe_bpr_users.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gorse_amd import capi, synth
from oracle import oracle as orc

L = capi.lib()
EPOCHS = 8
PER_SAMPLE, USERS, NO_REPLICAS, SORT_ON_UPDATE_STREAM = 1 << 28, 128, 32, 1 << 27
variants = [(PER_SAMPLE, "per-sample groups + replicas"), (PER_SAMPLE | NO_REPLICAS, "per-sample groups, no replicas"),
            (USERS, "user runs, sort under the update"), (USERS | (1 << 25), "user runs, negatives NOT through the replicas"),
            (USERS | SORT_ON_UPDATE_STREAM, "user runs, sort between updates"), (USERS | NO_REPLICAS, "user runs, no replicas")]
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
cases = [("ml1m", lambda: synth.s_ml1m(), 64), ("ml100k", lambda: synth.s_ml100k(), 16)]
if len(sys.argv) > 1 and sys.argv[1] == "c3":
    cases = []
if not quick:
    cases.append(("c3/8", lambda: synth.s_big_shard(rank=0, world=8), 128))
o = orc.Oracle()
for name, mk, d in cases:
    data = mk()
    U, I = data.U, data.I
    mf = capi.MF(U, I, d, data.uptr, data.uidx)
    P, Q = synth.init_factors(U, I, d, 0, 0.001, 1)
    has_test = data.test_idx.size > 0
    if has_test:
        rp, rq = P.copy(), Q.copy()
        srt = orc.sort_rows(data.uptr, data.uidx)
        t0 = time.perf_counter()
        for ep in range(EPOCHS):
            o.bpr_epoch_sampled(rp, rq, data.uptr, data.uidx, srt, 1, 10 + ep, 0, data.n_train, 0.05, 0.01)
        ref = o.evaluate(rp, rq, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0]
        print("%-6s sequential CPU oracle: NDCG@10 %.4f after %d epochs (%.1f s)" % (name, ref, EPOCHS, time.perf_counter() - t0),
              flush=True)
    for v, label in variants:
        L.gorse_hip_test_set_variant(v)
        mf.set_factors(P, Q)
        mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 9)  # warm-up: buffers, code objects
        mf.synchronize()
        mf.set_factors(P, Q)
        mf.set_profiling(True)
        mf.reset_profile()
        t0 = time.perf_counter()
        for e in range(EPOCHS):
            mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 10 + e)
        mf.synchronize()
        wall = time.perf_counter() - t0
        n, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
        ns, mss = mf.get_profile(capi.PROF_BPR_SORT)
        nq, msq = mf.get_profile(capi.PROF_BPR_SAMPLE)
        mf.set_profiling(False)
        gp, gq = mf.get_factors()
        ok = bool(np.isfinite(gp).all() and np.isfinite(gq).all())
        ndcg = float("nan")
        if has_test and ok:
            ndcg = o.evaluate(gp, gq, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0]
        print("%-6s d=%3d %-30s update %.3f ms/epoch (%.3e samples/s) sort %.3f sampler %.3f wall/epoch %.3f ms "
              "(%.3e samples/s) finite=%s NDCG %.4f" % (name, d, label, ms / EPOCHS, EPOCHS * data.n_train / (ms * 1e-3),
                                                       mss / EPOCHS, msq / EPOCHS, wall / EPOCHS * 1e3,
                                                       EPOCHS * data.n_train / wall, ok, ndcg), flush=True)
    L.gorse_hip_test_set_variant(0)
    mf.close()
