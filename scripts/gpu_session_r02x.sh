#!/bin/bash
# Round 2: the ALS row kernel with its waves in lockstep (accumulate together, solve together) against free-running.
set -u
TAG=${1:-r02_x}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 400 python scripts/gpu_probe_als.py phased > "$OUT/${TAG}_probe_als_phased.txt" 2>&1
echo "als probe exit $?"; cut -c1-330 "$OUT/${TAG}_probe_als_phased.txt"
