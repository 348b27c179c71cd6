#!/usr/bin/env python3
"""The symmetric sweep on the warm-start test's inputs (rows sorted by norm: a biased systematic sample), per metric: re-sweeps and
the symmetric search's counters next to the square sweep's re-sweeps."""
import sys

import numpy as np

sys.path.insert(0, ".")
from gorse_amd import capi  # noqa: E402

WARM_ALWAYS, SYM_OFF = 512, 1 << 23
N, d, k = 40000, 128, 100
rng = np.random.default_rng(N + d)
Xf = rng.standard_normal((N, d)).astype(np.float32)
Xf *= np.sort(rng.uniform(0.5, 2.0, N)).astype(np.float32)[:, None]
Xf[100:140] = Xf[5]
Xb = (np.ascontiguousarray(Xf).view(np.uint32) >> 16).astype(np.uint16)
capi.lib().gorse_hip_test_set_topk_path(2)
for metric, name in ((capi.METRIC_COSINE, "cosine"), (capi.METRIC_NEG_DOT, "-dot"), (capi.METRIC_EUCLIDEAN, "euclidean")):
    t = capi.TopK(Xb, metric, dtype=capi.DTYPE_BF16)
    for v, label in ((WARM_ALWAYS | SYM_OFF, "square"), (WARM_ALWAYS, "symmetric")):
        capi.lib().gorse_hip_test_set_topk_variant(v)
        t.all_pairs(k, fetch=False)
        print("%-10s %-10s symmetric %s: re-sweeps %d, tie/scan %s, stats [no pilot threshold, unverified, foreign overflow, staging overflow] %s"
              % (name, label, t.last_symmetric(), t.resweeps(), t.last_stats(), t.sym_stats()), flush=True)
        if v & SYM_OFF:
            capi.lib().gorse_hip_test_set_topk_variant(v | (1 << 24))
            t.all_pairs(k, fetch=False)
            fl, cn = t.pilot_state(N)
            print("   pilot alone: flags 0/1/2: %s; list lengths of the flagged: %s; of the others: min %d median %d max %d"
                  % (np.bincount(fl, minlength=3).tolist(), np.unique(cn[fl != 0])[:10].tolist(), cn[fl == 0].min(), np.median(cn[fl == 0]), cn[fl == 0].max()), flush=True)
            ft = t.warm_thresholds(N)
            print("   thresholds of the flagged: %s, unset without a flag: %d" % (np.unique(ft[fl != 0])[:5].tolist(), int((np.isinf(ft) & (fl == 0)).sum())), flush=True)
            capi.lib().gorse_hip_test_set_topk_variant(v)
            t.all_pairs(k, fetch=False)
            f = t.warm_thresholds(N)
            unset = np.isinf(f)
            print("   queries without a pilot threshold: %d; by index decile: %s; thresholds min %.3f max %.3f"
                  % (unset.sum(), [int(unset[i * N // 10:(i + 1) * N // 10].sum()) for i in range(10)], f[~unset].min(), f[~unset].max()), flush=True)
