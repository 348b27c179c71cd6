#!/usr/bin/env python3
"""S-ml1m BPR epochs at nFactors d (default 8) enqueued back to back, for a kernel timeline of one epoch (the `fit` leg of bench.py)."""
import sys

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 8
data = synth.s_ml1m()
P, Q = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 1)
mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx, data.iptr, data.iidx)
mf.set_factors(P, Q)
for ep in range(1, 31):
    mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 77, ep)
    if ep % 2 == 0:
        mf.epoch_throttle(2)
mf.synchronize() if hasattr(mf, "synchronize") else mf.get_factors()
print(mf.epoch_times(reset=False) if hasattr(mf, "epoch_times") else "")
