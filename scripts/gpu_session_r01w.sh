#!/bin/bash
set -u
TAG=${1:-r01_w}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
timeout 40 python -m pytest tests/test_gpu_vectors_db.py -q > "$OUT/${TAG}_pytest_vectors.log" 2>&1
echo "pytest exit $?"; tail -6 "$OUT/${TAG}_pytest_vectors.log" | cut -c1-250
