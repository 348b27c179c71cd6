#!/usr/bin/env python3
"""ALS epochs at the reference's own test shape (S-ml1m, model_test.go:93-104) for a kernel timeline, or against the row plan:
usage gpu_probe_als_small.py <nFactors> [epochs] [plan]   (plan: several (long row, chunk) thresholds, results compared with the default's)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 8
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
data = synth.s_ml1m()
P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.1, 3)
import numpy as np  # noqa: E402

plans = [(0, 0)] if len(sys.argv) <= 3 else [(0, 0), (2048, 2048), (1024, 1024), (512, 512), (256, 256), (1024, 256)]
ref = None
for (lr, ch) in plans:
    capi.lib().gorse_hip_test_set_als_plan(lr, ch)
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx, data.iptr, data.iidx)
    mf.set_factors(P0, Q0)
    mf.als_epoch(0.001, 0.06)
    t0 = time.perf_counter()
    for _ in range(epochs):
        mf.als_epoch(0.001, 0.06)
    dt = (time.perf_counter() - t0) / epochs * 1e3
    gP, gQ = mf.get_factors()
    if ref is None:
        ref = (gP, gQ)
    err = max(np.abs(gP - ref[0]).max() / np.abs(ref[0]).max(), np.abs(gQ - ref[1]).max() / np.abs(ref[1]).max())
    plan = "long row > %d in chunks of %d" % (lr, ch) if lr else "the library's plan (by the side's size)"
    print("S-ml1m ALS nFactors %d, %s: %.3f ms per epoch over %d epochs (synchronous calls); max |diff| / max |ref| to the library's plan %.2e"
          % (d, plan, dt, epochs, err), flush=True)
    mf.close()
capi.lib().gorse_hip_test_set_als_plan(0, 0)
