#!/usr/bin/env python3
"""GPU probe: attribute the Hogwild update kernel's time (loads vs the three atomic row updates)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from gorse_amd import capi, synth

names = {0: "base", 1: "plain loads", 2: "no P writes", 4: "no Qi writes", 8: "no Qj writes", 14: "no writes",
         15: "no writes, plain loads", 6: "only Qj writes", 10: "only Qi writes", 12: "only P writes"}
for name, U, I, N, d in [("ml1m", 6040, 3706, 994169, 64), ("mid", 125000, 200000, 4000000, 128)]:
    for z in (1.0, 0.0):
        data = synth.synth_cf(U, I, N, seed=42, zipf_s=z, min_len=1 if name == "mid" else 19, with_test=False)
        mf = capi.MF(U, I, d, data.uptr, data.uidx)
        P, Q = synth.init_factors(U, I, d, 0, 0.001, 1)
        for v in (0, 1, 2, 4, 8, 14, 15, 6, 10, 12):
            capi.lib().gorse_hip_test_set_variant(v)
            mf.set_factors(P, Q)
            mf.bpr_epoch(data.n_train, 0.05, 0.01, 1, 0)
            mf.set_profiling(True)
            mf.reset_profile()
            for e in range(3):
                mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 10 + e)
            mf.synchronize()
            n, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
            mf.set_profiling(False)
            print("%-5s d=%3d zipf=%.1f %-24s %.3f ms/launch %.3e samples/s" % (name, d, z, names[v], ms / n,
                                                                                3 * data.n_train / (ms * 1e-3)), flush=True)
        capi.lib().gorse_hip_test_set_variant(0)
        mf.close()
