#!/usr/bin/env python3
"""ALS epochs at the reference's own test shape (S-ml1m, model_test.go:93-104) for a kernel timeline: usage gpu_probe_als_small.py <nFactors> [epochs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 8
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
data = synth.s_ml1m()
P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.1, 3)
mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx, data.iptr, data.iidx)
mf.set_factors(P0, Q0)
mf.als_epoch(0.001, 0.06)
t0 = time.perf_counter()
for _ in range(epochs):
    mf.als_epoch(0.001, 0.06)
print("S-ml1m ALS nFactors %d: %.3f ms per epoch over %d epochs (synchronous calls)" % (d, (time.perf_counter() - t0) / epochs * 1e3, epochs), flush=True)
