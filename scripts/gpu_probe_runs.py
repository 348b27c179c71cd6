#!/usr/bin/env python3
"""GPU probe: item-run BPR schedule vs the round-1 per-sample atomic kernel: run-block length, resident
workgroups (their product x 16 = span of sorted positions applied concurrently), sort window, write
ablations, sort/sampler cost, end-to-end epoch rate, and NDCG@10 after 8 epochs against the sequential
CPU oracle (ml1m / ml100k shapes).  Output -> profiles/rNN_*_probe_bpr_runs.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from gorse_amd import capi, synth
from oracle import oracle as orc

L = capi.lib()
EPOCHS = 8


def V(block=0, window=0, wgs=0, abl=0):
    b = {0: 0, 1: 1, 2: 2, 4: 3, 8: 4, 16: 5, 32: 6}[block]
    return abl | (b << 8) | (window << 12) | (wgs << 20)


RUNS = 64
variants = [(32, "per-sample atomics (round 1)"), (0, "per-sample + hot-row replicas"),
            (RUNS, "runs blk4 x512WG (span 32K)"), (RUNS | V(block=8, wgs=8), "runs blk8 x256WG (32K)"),
            (RUNS | V(block=16, wgs=7), "runs blk16 x128WG (32K)"),
            (2, "replicas, no P writes"), (8, "replicas, no Qj writes"), (4, "replicas, no Qi writes"),
            (14, "replicas, no writes"), (1, "replicas, plain loads")]
cases = [("ml1m", lambda: synth.s_ml1m(), 64), ("ml100k", lambda: synth.s_ml100k(), 16),
         ("mid", lambda: synth.synth_cf(125000, 200000, 12500000, seed=42, zipf_s=1.0, min_len=1, with_test=False), 128),
         ("mid-u", lambda: synth.synth_cf(125000, 200000, 12500000, seed=42, zipf_s=0.0, min_len=1, with_test=False), 128)]
o = orc.Oracle()
for name, mk, d in cases:
    data = mk()
    U, I = data.U, data.I
    top = np.bincount(data.uidx, minlength=I).max() / data.n_train
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
        print("%-6s sequential CPU oracle: NDCG@10 %.4f after %d epochs (%.1f s)" % (name, ref, EPOCHS,
                                                                                      time.perf_counter() - t0), flush=True)
    for v, label in variants:
        L.gorse_hip_test_set_variant(v)
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
        if has_test and ok and not (v & 14):
            ndcg = o.evaluate(gp, gq, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0]
        print("%-6s d=%3d top=%.4f %-30s update %.3f ms/epoch (%.3e samples/s) sort %.3f sampler %.3f "
              "wall/epoch %.3f ms (%.3e samples/s) finite=%s NDCG %.4f"
              % (name, d, top, label, ms / EPOCHS, EPOCHS * data.n_train / (ms * 1e-3), mss / EPOCHS, msq / EPOCHS,
                 wall / EPOCHS * 1e3, EPOCHS * data.n_train / wall, ok, ndcg), flush=True)
    L.gorse_hip_test_set_variant(0)
    mf.close()
