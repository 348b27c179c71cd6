#!/usr/bin/env python3
"""GPU probe: C2 epochs (update kernel ms, epoch ms) with whatever library GORSE_HIP_LIB names -- used to compare builds of csrc/bpr.hip
with a different number of replica rows per hot item (-DGORSE_HOT_REPLICAS=16: 0.627 ms either way, round 4)."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from gorse_amd import capi, synth
data = synth.s_ml1m()
d = 64
P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 1)
mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
mf.set_factors(P0, Q0)
for r in range(3):
    mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 100 + r)
mf.synchronize()
mf.set_profiling(True); mf.reset_profile()
t0 = time.perf_counter()
E = 20
for e in range(E):
    mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 77, 1 + e)
mf.synchronize()
wall = (time.perf_counter() - t0) / E * 1e3
n, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
print(os.environ.get("GORSE_HIP_LIB", "default"), "update %.3f ms wall %.3f ms" % (ms / E, wall))
