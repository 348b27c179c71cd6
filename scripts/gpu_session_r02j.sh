#!/bin/bash
# Round 2, tenth device session: the whole GPU suite as the driver runs it, then sparse (8-deep look-ahead, groups of 2048),
# ALS (S in LDS, one 8-wave workgroup per CU) and the default bench line.
set -u
TAG=${1:-r02_j}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -q -m gpu -x ) > "$OUT/${TAG}_pytest_gpu.log" 2>&1
echo "pytest gpu exit $?"; tail -12 "$OUT/${TAG}_pytest_gpu.log"
timeout 300 python scripts/gpu_probe_sparse_trace.py c3 > "$OUT/${TAG}_probe_sparse_trace.txt" 2>&1
echo "sparse trace exit $?"; cut -c1-300 "$OUT/${TAG}_probe_sparse_trace.txt"
timeout 200 python scripts/gpu_probe_als.py prof > "$OUT/${TAG}_probe_als_prof.txt" 2>&1
echo "probe als prof exit $?"; cut -c1-460 "$OUT/${TAG}_probe_als_prof.txt"
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"
echo "bench default exit $?"; python - "$OUT/${TAG}_bench_default.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        for k in (None, "topk", "c3", "i2i", "als"):
            o = d if k is None else d.get(k, {})
            print(k or "main", o.get("value"), o.get("unit"), "ms/step", o.get("ms_per_step"), "frac", (o.get("roofline") or {}).get("frac"), o.get("error"),
                  "cpu", (o.get("cpu_baseline") or {}).get("value"))
PY
tail -4 "$OUT/${TAG}_bench_default.err"
