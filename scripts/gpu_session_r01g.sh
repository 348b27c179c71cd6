#!/bin/bash
set -u
TAG=${1:-r01_g}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 200 python scripts/gpu_probe_topk.py prof > "$OUT/${TAG}_probe_topk_prof.txt" 2>&1
echo "probe topk prof exit $?"; cat "$OUT/${TAG}_probe_topk_prof.txt"
