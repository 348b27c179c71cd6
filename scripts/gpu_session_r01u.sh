#!/bin/bash
set -u
TAG=${1:-r01_u}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
timeout 70 python -m pytest tests/test_gpu_topk_mfma.py tests/test_gpu_topk_sgemm.py -q -x > "$OUT/${TAG}_pytest_topk.log" 2>&1
echo "pytest exit $?"; tail -5 "$OUT/${TAG}_pytest_topk.log" | cut -c1-300
timeout 40 python - > "$OUT/${TAG}_probe_euclid.txt" 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np
from gorse_amd import capi
rng = np.random.default_rng(1)
N, d, k = 200_000, 64, 100
X = rng.standard_normal((N, d)).astype(np.float32)
for metric, name in ((capi.METRIC_EUCLIDEAN, "euclidean"), (capi.METRIC_NEG_DOT, "-dot")):
    t = capi.TopK(X, metric)
    t.all_pairs(k, 0, 4096, fetch=False)
    t0 = time.perf_counter()
    t.all_pairs(k, 0, N, fetch=False)
    t.synchronize()
    dt = time.perf_counter() - t0
    print("fp32 %dx%d %s top-%d all pairs: %.1f ms (%.3e pairs/s), scan fallback %d, tie-replayed %d"
          % (N, d, name, k, dt * 1e3, N * (N - 1) / dt, *t.last_stats()), flush=True)
PY
cat "$OUT/${TAG}_probe_euclid.txt"
