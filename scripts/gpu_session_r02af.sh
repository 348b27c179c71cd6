#!/bin/bash
# Round 2, ranking buffers of 2 x 128 keys for k <= 128 (twelve waves per CU instead of ten).
set -u
TAG=${1:-r02_af}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_vectors_sparse.py tests/test_gpu_vectors_db.py -q -m gpu -x > "$OUT/${TAG}_pytest.log" 2>&1
echo "pytest exit $?"; tail -5 "$OUT/${TAG}_pytest.log"
timeout 300 python scripts/gpu_probe_sparse_trace.py c3 0 1 > "$OUT/${TAG}_probe_sparse_trace.txt" 2>&1
echo "sparse trace exit $?"; cut -c1-400 "$OUT/${TAG}_probe_sparse_trace.txt"
timeout 300 python bench.py --workload i2i --steps 3 --warmup 1 > "$OUT/${TAG}_bench_i2i.json" 2> "$OUT/${TAG}_bench_i2i.err"
echo "bench i2i exit $?"; python - "$OUT/${TAG}_bench_i2i.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d.get("cpu_baseline", {}).get("value"))
PY
