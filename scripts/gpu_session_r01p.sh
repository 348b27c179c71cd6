#!/bin/bash
set -u
TAG=${1:-r01_p}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_cf_parity.py -q > "$OUT/${TAG}_pytest_cf.log" 2>&1
echo "pytest cf exit $?"; tail -3 "$OUT/${TAG}_pytest_cf.log"
timeout 200 python scripts/gpu_probe_users.py > "$OUT/${TAG}_probe_bpr_users.txt" 2>&1
echo "probe users exit $?"; grep -v "no replicas" "$OUT/${TAG}_probe_bpr_users.txt"
timeout 200 python scripts/gpu_probe_als.py prof > "$OUT/${TAG}_probe_als_prof.txt" 2>&1
echo "probe als prof exit $?"; grep "without\|user rows" "$OUT/${TAG}_probe_als_prof.txt"
