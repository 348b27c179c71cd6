#!/bin/bash
set -u
TAG=${1:-r01_s}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
timeout 60 python -m pytest tests/test_gpu_cf_parity.py -q -k "evaluate_on_device or row_sharded or score or rank" > "$OUT/${TAG}_pytest_eval.log" 2>&1
echo "pytest exit $?"; tail -4 "$OUT/${TAG}_pytest_eval.log"
