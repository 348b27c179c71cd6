#!/usr/bin/env python3
"""GPU probe: does spreading the positive-item atomics of the per-sample kernel over R replica rows
remove the popularity slowdown?  (speed only: reads still use the base rows)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from gorse_amd import capi, synth

L = capi.lib()
for name, mk, d in [("ml1m", lambda: synth.s_ml1m(), 64),
                    ("mid", lambda: synth.synth_cf(125000, 200000, 12500000, seed=42, zipf_s=1.0, min_len=1, with_test=False), 128)]:
    data = mk()
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
    P, Q = synth.init_factors(data.U, data.I, d, 0, 0.001, 1)
    for v, label in [(16, "per-sample atomics"), (16 | 32, "Qi -> replica R=1 (separate buffer)"),
                     (16 | 32 | (1 << 24), "R=2"), (16 | 32 | (2 << 24), "R=4"), (16 | 32 | (3 << 24), "R=8"),
                     (16 | 32 | (4 << 24), "R=16"), (16 | 32 | (6 << 24), "R=64"), (16 | 4, "no Qi writes"),
                     (16 | 1, "plain loads"), (16 | 1 | 32 | (3 << 24), "plain loads + R=8")]:
        L.gorse_hip_test_set_variant(v)
        mf.set_factors(P, Q)
        mf.bpr_epoch(data.n_train, 0.05, 0.01, 1, 0)
        mf.set_profiling(True)
        mf.reset_profile()
        for e in range(4):
            mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 10 + e)
        mf.synchronize()
        n, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
        mf.set_profiling(False)
        print("%-5s d=%3d %-40s %.3f ms/epoch %.3e samples/s" % (name, d, label, ms / 4, 4 * data.n_train / (ms * 1e-3)),
              flush=True)
    L.gorse_hip_test_set_variant(0)
    mf.close()
