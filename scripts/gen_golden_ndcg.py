#!/usr/bin/env python3
"""Golden values for the full-size NDCG gates of tests/test_gpu_baseline_configs.py: NDCG@10 of 8192 held-out users after ONE
sequential (Jobs = 1) BPR epoch of the oracle, nFactors 128, over
  c3full : the WHOLE S-big set (1M x 200K x 100M)                    -> tests/golden/c3full_oracle_ndcg.json
  big    : north_star's 10M x 1M set at 250M draws (220M feedbacks)  -> tests/golden/big_oracle_ndcg.json
-- 1e8 resp. 2.2e8 sequential SGD steps, minutes to a quarter of an hour of one host core, which is why the GPU tests read the
number from a committed file instead of recomputing it on the GPU box.  CPU only; run from the repo root:
    python scripts/gen_golden_ndcg.py c3full|big
The inputs are the seeded generators of gorse_amd/synth.py (numpy PCG64), identical here and on the GPU box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gorse_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HELD, NEG, HSEED, D, LR, REG, SEED = 8192, 99, 5, 128, 0.05, 0.01, 77
BIG_DRAWS = 250_000_000  # bench.py's `big` leg and test_big_10m_users_d128_one_epoch_ndcg use the same figure


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c3full"
    o = orc.Oracle()
    t0 = time.perf_counter()
    if which == "big":
        full, name, init = synth.s_huge(N=BIG_DRAWS), "S-huge (10M x 1M, %d draws)" % BIG_DRAWS, synth.init_factors_big
    else:
        full, name, init = synth.s_big_full(), "S-big whole", synth.init_factors
    data = synth.hold_out(full, HELD, NEG, HSEED)
    print("data: %d users, %d items, %d train feedbacks (%.0f s)" % (data.U, data.I, data.n_train, time.perf_counter() - t0), flush=True)
    P, Q = init(data.U, data.I, D, 0.0, 0.001, 1)
    ev = lambda: float(o.evaluate(P, Q, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0])
    base = ev()
    srt = orc.sort_rows(data.uptr, data.uidx)
    t0 = time.perf_counter()
    o.bpr_epoch_sampled(P, Q, data.uptr, data.uidx, srt, SEED, 1, 0, data.n_train, LR, REG)
    dt = time.perf_counter() - t0
    out = {"what": "NDCG@10 of the first %d users with >= 2 feedbacks (leave-one-out, %d negatives, hold_out seed %d) after one "
                   "sequential oracle epoch (orc_bpr_epoch_sampled, seed %d, epoch 1) over %s, nFactors %d, lr %g, reg %g, "
                   "init N(0, 0.001) seed 1" % (HELD, NEG, HSEED, SEED, name, D, LR, REG),
           "n_train": data.n_train, "ndcg_untrained": base, "ndcg_after_one_epoch": ev(), "oracle_epoch_seconds": dt,
           "numpy": np.__version__, "generator": "scripts/gen_golden_ndcg.py " + which}
    print(json.dumps(out, indent=1))
    with open(os.path.join(ROOT, "tests", "golden", "%s_oracle_ndcg.json" % which), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
