#!/usr/bin/env python3
"""BPR epochs with the preparation ON THE UPDATE STREAM (variant bit 22) next to the production schedule (preparation of chunk c + 1
under the update of chunk c), for a kernel timeline under rocprofv3: the serial runs show every kernel's duration alone.
usage: gpu_probe_bpr_serial.py <shape: ml1m | ml100k | c3s> <serial | overlapped | both> <nFactors> [<nFactors> ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "ml1m"
which = sys.argv[2] if len(sys.argv) > 2 else "both"
widths = [int(x) for x in sys.argv[3:]] or [16]
data = synth.s_ml1m() if shape == "ml1m" else (synth.s_ml100k() if shape == "ml100k" else synth.s_big_shard())
epochs = 3 if shape == "c3s" else 8
for d in widths:
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 3)
    for variant, name in ((0, "overlapped"), (1 << 22, "serial")):
        if which not in ("both", name):
            continue
        capi.lib().gorse_hip_test_set_variant(variant)
        mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
        mf.set_factors(P0, Q0)
        mf.bpr_epoch(data.n_train, 0.05, 0.01, 7, 1, mode=capi.BPR_HOGWILD_STORES)
        t0 = time.perf_counter()
        for ep in range(1, epochs + 1):
            mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 7, ep, mode=capi.BPR_HOGWILD_STORES)
        mf.synchronize()
        print("%s nFactors %d %s: %.3f ms per epoch over %d enqueued epochs" % (shape, d, name, (time.perf_counter() - t0) / epochs * 1e3, epochs),
              flush=True)
        del mf
capi.lib().gorse_hip_test_set_variant(0)
