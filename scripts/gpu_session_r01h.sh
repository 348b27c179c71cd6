#!/bin/bash
set -u
TAG=${1:-r01_h}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_topk_mfma.py tests/test_gpu_topk_sgemm.py -q > "$OUT/${TAG}_pytest_topk.log" 2>&1
echo "pytest topk exit $?"; tail -4 "$OUT/${TAG}_pytest_topk.log"
timeout 200 python scripts/gpu_probe_topk.py prof > "$OUT/${TAG}_probe_topk_prof.txt" 2>&1
echo "probe topk prof exit $?"; cat "$OUT/${TAG}_probe_topk_prof.txt"
timeout 200 python scripts/gpu_probe_topk.py variants > "$OUT/${TAG}_probe_topk_variants.txt" 2>&1
echo "probe topk variants exit $?"; cat "$OUT/${TAG}_probe_topk_variants.txt"
