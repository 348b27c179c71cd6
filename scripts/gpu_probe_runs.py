#!/usr/bin/env python3
"""GPU probe: item-run BPR schedule vs the round-1 per-sample atomic kernel, run-block length sweep,
write ablations, sort/sampler cost and end-to-end epoch rate.  Output -> profiles/rNN_*_probe_bpr_runs.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from gorse_amd import capi, synth

L = capi.lib()
cases = [("ml1m", 6040, 3706, 994169, 64, 19), ("ml100k", 943, 1682, 99057, 16, 19),
         ("mid", 125000, 200000, 12500000, 128, 1)]
variants = [(16, "round-1 per-sample atomics"), (0, "item runs, auto block"), (4 << 8, "item runs, block 16"),
            (5 << 8, "item runs, block 32"), (6 << 8, "item runs, block 64"), (7 << 8, "item runs, block 128"),
            (2, "runs, no P writes"), (8, "runs, no Qj writes"), (4, "runs, no Qi flush"), (14, "runs, no writes"),
            (1, "runs, plain loads")]
for name, U, I, N, d, min_len in cases:
    for z in (1.0, 0.0):
        if name == "ml100k" and z == 0.0:
            continue
        data = synth.synth_cf(U, I, N, seed=42, zipf_s=z, min_len=min_len, with_test=False)
        top = np.bincount(data.uidx, minlength=I).max() / data.n_train
        mf = capi.MF(U, I, d, data.uptr, data.uidx)
        P, Q = synth.init_factors(U, I, d, 0, 0.001, 1)
        for v, label in variants:
            L.gorse_hip_test_set_variant(v)
            mf.set_factors(P, Q)
            mf.bpr_epoch(data.n_train, 0.05, 0.01, 1, 0)
            mf.set_profiling(True)
            mf.reset_profile()
            reps = 5
            t0 = time.perf_counter()
            for e in range(reps):
                mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 10 + e)
            mf.synchronize()
            wall = time.perf_counter() - t0
            n, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
            ns, mss = mf.get_profile(capi.PROF_BPR_SORT)
            nq, msq = mf.get_profile(capi.PROF_BPR_SAMPLE)
            mf.set_profiling(False)
            gp, gq = mf.get_factors()
            ok = bool(np.isfinite(gp).all() and np.isfinite(gq).all())
            print("%-6s d=%3d zipf=%.1f top=%.4f %-28s update %.3f ms (%.3e samples/s) sort %.3f ms sampler %.3f ms "
                  "wall/epoch %.3f ms (%.3e samples/s) finite=%s"
                  % (name, d, z, top, label, ms / n, reps * data.n_train / (ms * 1e-3), mss / max(ns, 1),
                     msq / max(nq, 1), wall / reps * 1e3, reps * data.n_train / wall, ok), flush=True)
        L.gorse_hip_test_set_variant(0)
        mf.close()
