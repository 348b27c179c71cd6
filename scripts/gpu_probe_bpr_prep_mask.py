#!/usr/bin/env python3
"""The preparation stream confined to every n-th CU (hipExtStreamCreateWithCUMask): ms per enqueued epoch at S-ml1m (nFactors 8, 16,
64) and at the C3 shard (nFactors 128), n = 1 (the whole chip), 2, 4, 8, 16."""
import sys
import time

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402

L = capi.lib()


def run(data, d, epochs, label):
    row = []
    for n in (1, 2, 4, 8, 16):
        L.gorse_hip_test_set_prep_cu_stride(n)
        P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 3)
        mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
        mf.set_factors(P0, Q0)
        mf.bpr_epoch(data.n_train, 0.05, 0.01, 7, 1, mode=capi.BPR_HOGWILD_STORES)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for ep in range(1, epochs + 1):
                mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 7, ep, mode=capi.BPR_HOGWILD_STORES)
            mf.synchronize()
            best = min(best, (time.perf_counter() - t0) / epochs * 1e3)
        row.append("1/%d: %.3f" % (n, best))
        mf.close()
    L.gorse_hip_test_set_prep_cu_stride(0)
    print("%-28s ms per epoch, preparation stream on %s" % (label, "  ".join(row)), flush=True)


ml1m = synth.s_ml1m()
for d in (8, 16, 64):
    run(ml1m, d, 20, "S-ml1m nFactors %d" % d)
run(synth.s_big_shard(rank=0, world=8), 128, 5, "C3 shard nFactors 128")
