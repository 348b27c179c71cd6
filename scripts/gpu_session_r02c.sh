#!/bin/bash
# Round 2, third device session: sparse top-k with the 64-lists-at-once path, the RCCL entry points at world 1, the full
# default bench line (C2 + C4 + c3 + i2i + als objects).
set -u
TAG=${1:-r02_c}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_vectors_sparse.py tests/test_gpu_comm.py -q -m gpu -x > "$OUT/${TAG}_pytest_sparse_comm.log" 2>&1
echo "pytest sparse + comm exit $?"; tail -25 "$OUT/${TAG}_pytest_sparse_comm.log"
timeout 400 python scripts/gpu_probe_sparse.py c3 > "$OUT/${TAG}_probe_sparse_c3.txt" 2>&1
echo "probe sparse c3 exit $?"; cut -c1-330 "$OUT/${TAG}_probe_sparse_c3.txt"
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
