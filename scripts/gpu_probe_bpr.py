#!/usr/bin/env python3
"""GPU probe: BPR update-kernel throughput by schedule and nFactors (not part of the test suite)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from gorse_amd import capi, synth

cases = [("ml1m", 6040, 3706, 994169), ("mid", 125000, 200000, 4000000)]
for name, U, I, N in cases:
    data = synth.synth_cf(U, I, N, seed=42, min_len=1 if name == "mid" else 19, with_test=False)
    for d in (16, 64, 128):
        for mode in (capi.BPR_HOGWILD_ATOMIC, capi.BPR_HOGWILD_RACY):
            mf = capi.MF(U, I, d, data.uptr, data.uidx)
            P, Q = synth.init_factors(U, I, d, 0, 0.001, 1)
            mf.set_factors(P, Q)
            for w in range(2):
                mf.bpr_epoch(data.n_train, 0.05, 0.01, 1, w, mode=mode)
            mf.set_profiling(True)
            mf.reset_profile()
            t0 = time.perf_counter()
            steps = 5
            for s in range(steps):
                mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 10 + s, mode=mode)
            mf.synchronize()
            dt = time.perf_counter() - t0
            n, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
            ns, mss = mf.get_profile(capi.PROF_BPR_SAMPLE)
            bps = 6 * d * 4 + 12
            sps = steps * data.n_train / dt
            ksps = steps * data.n_train / (ms * 1e-3)
            print("%-5s d=%3d mode=%d wall %.3e samples/s | update kernel %.3f ms/launch -> %.3e samples/s = %.0f GB/s alg | sampler %.3f ms"
                  % (name, d, mode, sps, ms / n, ksps, ksps * bps / 1e9, mss / max(ns, 1)), flush=True)
            mf.close()
