#!/bin/bash
# Round 2, last session: the whole GPU suite and the default bench line on the final library.
set -u
TAG=${1:-r02_ao}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -s -m gpu > "$OUT/${TAG}_pytest_gpu.log" 2>&1
echo "pytest gpu exit $?"; grep -n "passed\|failed" "$OUT/${TAG}_pytest_gpu.log" | tail -2; grep -n "^FAILED" "$OUT/${TAG}_pytest_gpu.log" | head
timeout 400 python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"
echo "bench exit $?"; python - "$OUT/${TAG}_bench_default.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print("C2", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"])
        for k in ("topk", "c3", "i2i", "als"):
            if k in d and "value" in d[k]:
                print(k, d[k]["value"], d[k]["ms_per_step"], d[k]["roofline"]["frac"], d[k]["roofline"]["traffic"])
            elif k in d:
                print(k, d[k])
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
