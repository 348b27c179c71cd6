#!/bin/bash
# Round 2: the heavy-query kernel with one work queue per XCD (one dense vector per L2 at a time).
set -u
TAG=${1:-r02_ai}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_vectors_sparse.py tests/test_gpu_vectors_db.py -q -m gpu -x > "$OUT/${TAG}_pytest.log" 2>&1
echo "pytest exit $?"; tail -3 "$OUT/${TAG}_pytest.log"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}" -o bench -- python "$ROOT/bench.py" --workload i2i --steps 3 --warmup 1 > "$OUT/${TAG}_bench_i2i.json" 2> "$OUT/${TAG}_bench_i2i.err"
python "$ROOT/scripts/rocpd_summary.py" "$(find "$OUT/prof_${TAG}" -name '*_results.db' | head -1)" > "$OUT/${TAG}_kernel_stats_i2i.txt" 2>&1
head -8 "$OUT/${TAG}_kernel_stats_i2i.txt" | cut -c1-170
cd "$ROOT"
python - "$OUT/${TAG}_bench_i2i.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
rm -rf "$OUT"/prof_${TAG}
