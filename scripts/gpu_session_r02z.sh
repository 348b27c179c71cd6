#!/bin/bash
# Round 2: the ALS Gram accumulation with three gather stages in flight.
set -u
TAG=${1:-r02_z}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_cf_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_host_mirror.py -q -m gpu -x -k "als or ALS" > "$OUT/${TAG}_pytest_als.log" 2>&1
echo "pytest als exit $?"; tail -3 "$OUT/${TAG}_pytest_als.log"
timeout 400 python scripts/gpu_probe_als.py prof > "$OUT/${TAG}_probe_als_prof.txt" 2>&1
echo "als probe exit $?"; cut -c1-330 "$OUT/${TAG}_probe_als_prof.txt"
timeout 300 python bench.py --workload als --steps 4 --warmup 1 > "$OUT/${TAG}_bench_als.json" 2> "$OUT/${TAG}_bench_als.err"
echo "bench als exit $?"; python - "$OUT/${TAG}_bench_als.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
