#!/usr/bin/env python3
"""GPU probe: where the time of the C2 epoch (S-ml1m, 6040 user runs) goes -- update-kernel ms against the number of samples of the
epoch (fixed cost against per-sample cost), with and without the hot-row replicas + folder workgroups, at d = 64 and 16.

usage: gpu_probe_bpr_c2_scaling.py      Output -> profiles/rNN_*_probe_bpr_c2_scaling.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from gorse_amd import capi, synth

L = capi.lib()
data = synth.s_ml1m()
for d in (64, 16):
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 1)
    for variant, label in ((128, "replicas"), (128 | 32, "no replicas")):
        L.gorse_hip_test_set_variant(variant)
        mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
        for n in (62_500, 125_000, 250_000, 500_000, 994_169, 2_000_000, 4_000_000):
            mf.set_factors(P0, Q0)
            mf.bpr_epoch_enqueue(n, 0.05, 0.01, 1, 9)
            mf.synchronize()
            mf.set_profiling(True)
            mf.reset_profile()
            reps = 8
            for e in range(reps):
                mf.bpr_epoch_enqueue(n, 0.05, 0.01, 77, 1 + e)
            mf.synchronize()
            _, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
            mf.set_profiling(False)
            print("d=%3d %-12s n=%8d (%.0f per user) update %8.4f ms/epoch  %.3f us per sample of a run  %.3e samples/s" % (
                d, label, n, n / data.U, ms / reps, ms / reps * 1e3 / (n / data.U), reps * n / (ms * 1e-3)), flush=True)
        mf.close()
L.gorse_hip_test_set_variant(0)
