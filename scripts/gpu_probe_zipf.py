#!/usr/bin/env python3
"""GPU probe: how much of the BPR update kernel's time is hot-row serialisation? (Zipf exponent sweep)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from gorse_amd import capi, synth

for name, U, I, N, d in [("ml1m", 6040, 3706, 994169, 64), ("mid", 125000, 200000, 4000000, 128)]:
    for s in (0.0, 0.5, 1.0):
        data = synth.synth_cf(U, I, N, seed=42, zipf_s=s, min_len=1 if name == "mid" else 19, with_test=False)
        top = np.bincount(data.uidx, minlength=I).max() / data.n_train
        mf = capi.MF(U, I, d, data.uptr, data.uidx)
        P, Q = synth.init_factors(U, I, d, 0, 0.001, 1)
        mf.set_factors(P, Q)
        mf.bpr_epoch(data.n_train, 0.05, 0.01, 1, 0)
        mf.set_profiling(True)
        mf.reset_profile()
        for e in range(5):
            mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 10 + e)
        mf.synchronize()
        n, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
        ksps = 5 * data.n_train / (ms * 1e-3)
        print("%-5s d=%3d zipf=%.1f top-item share %.4f : update %.3f ms/launch %.3e samples/s %.0f GB/s alg"
              % (name, d, s, top, ms / n, ksps, ksps * (6 * d * 4 + 12) / 1e9), flush=True)
        mf.close()
