#!/bin/bash
set -u
TAG=${1:-r01_n}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_cf_parity.py -q -k als > "$OUT/${TAG}_pytest_als.log" 2>&1
echo "pytest als exit $?"; tail -3 "$OUT/${TAG}_pytest_als.log"
timeout 200 python scripts/gpu_probe_als.py prof > "$OUT/${TAG}_probe_als_prof.txt" 2>&1
echo "probe als prof exit $?"; cat "$OUT/${TAG}_probe_als_prof.txt"
