#!/usr/bin/env python3
"""GPU probe: BPR epochs with the update stream at a higher stream priority than the sampler / sort stream, against equal
priorities (gorse_hip_test_set_stream_priorities): wall time per epoch at C2 and at the C3 shard."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402

L = capi.lib()
for name, mk, d, epochs in (("ml1m", synth.s_ml1m, 64, 40), ("c3/8", lambda: synth.s_big_shard(rank=0, world=8), 128, 6)):
    data = mk()
    P, Q = synth.init_factors(data.U, data.I, d, 0, 0.001, 1)
    for on in (1, 0, 1, 0):
        L.gorse_hip_test_set_stream_priorities(on)
        mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
        mf.set_factors(P, Q)
        for e in range(3):
            mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, e + 1)
        mf.synchronize()
        t0 = time.perf_counter()
        for e in range(epochs):
            mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 10 + e)
        mf.synchronize()
        dt = (time.perf_counter() - t0) / epochs
        print("%-5s d=%3d update stream %s: %.4f ms per epoch (%.3e samples/s)" % (name, d, "ahead" if on else "equal", dt * 1e3, data.n_train / dt), flush=True)
        mf.close()
L.gorse_hip_test_set_stream_priorities(0)
