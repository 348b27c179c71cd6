#!/bin/bash
set -u
TAG=${1:-r01_v}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
timeout 75 python -m pytest tests -q -m gpu > "$OUT/${TAG}_pytest_gpu.log" 2>&1
echo "pytest exit $?"; tail -5 "$OUT/${TAG}_pytest_gpu.log" | cut -c1-300
