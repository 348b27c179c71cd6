#!/bin/bash
# Round 2: the Gram form for 64 < nFactors <= 128 (als_wide_kernel).
set -u
TAG=${1:-r02_an}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_cf_parity.py tests/test_gpu_comm.py -q -s -m gpu -x -k "als or ALS or library_communicator" > "$OUT/${TAG}_pytest_als.log" 2>&1
echo "pytest als exit $?"; grep -n "passed\|failed" "$OUT/${TAG}_pytest_als.log" | tail -2; grep -n "wide" "$OUT/${TAG}_pytest_als.log" | cut -c1-200 | head -12
timeout 240 python scripts/gpu_probe_als.py wide > "$OUT/${TAG}_probe_als_wide.txt" 2>&1
echo "als probe exit $?"; cut -c1-260 "$OUT/${TAG}_probe_als_wide.txt"
