#!/usr/bin/env python3
"""S-ml1m, one width, 12 enqueued epochs of the production schedule (for a kernel timeline under rocprofv3).  usage: gpu_probe_bpr_epochs.py d [ml1m | ml100k]"""
import sys
import time

sys.path.insert(0, ".")
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shape = sys.argv[2] if len(sys.argv) > 2 else "ml1m"
data = synth.s_ml100k() if shape == "ml100k" else synth.s_ml1m()
P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 3)
mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
mf.set_factors(P0, Q0)
mf.bpr_epoch(data.n_train, 0.05, 0.01, 7, 1, mode=capi.BPR_HOGWILD_STORES)
for rep in range(2):
    t0 = time.perf_counter()
    for ep in range(1, 13):
        mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 7, ep, mode=capi.BPR_HOGWILD_STORES)
    mf.synchronize()
    print("%s nFactors %d: %.3f ms per epoch over 12 enqueued epochs" % (shape, d, (time.perf_counter() - t0) / 12 * 1e3), flush=True)
