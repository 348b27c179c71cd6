#!/bin/bash
# One short gpurun session: tie-replay + ALS Gram-form parity tests, ALS and top-k probes.  Every step has its own timeout.
set -u
TAG=${1:-r01_d}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_gpu_topk_mfma.py -q -x > "$OUT/${TAG}_pytest_topk_mfma.log" 2>&1
echo "pytest topk_mfma exit $?"; tail -4 "$OUT/${TAG}_pytest_topk_mfma.log"
timeout 240 python -m pytest tests/test_gpu_cf_parity.py -q -k als > "$OUT/${TAG}_pytest_als.log" 2>&1
echo "pytest als exit $?"; tail -4 "$OUT/${TAG}_pytest_als.log"
timeout 200 python scripts/gpu_probe_als.py quick > "$OUT/${TAG}_probe_als.txt" 2>&1
echo "probe als exit $?"; tail -8 "$OUT/${TAG}_probe_als.txt"
timeout 150 python scripts/gpu_probe_topk.py c4 > "$OUT/${TAG}_probe_topk.txt" 2>&1
echo "probe topk exit $?"; tail -4 "$OUT/${TAG}_probe_topk.txt"
