#!/bin/bash
set -u
TAG=${1:-r01_t}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
timeout 80 python -m pytest tests/test_gpu_host_mirror.py -q -s > "$OUT/${TAG}_pytest_host.log" 2>&1
echo "pytest exit $?"; grep "NDCG\|passed\|failed" "$OUT/${TAG}_pytest_host.log" | cut -c1-200
