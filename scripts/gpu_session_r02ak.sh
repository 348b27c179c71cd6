#!/bin/bash
# Round 2: stream priorities of the BPR handle (update ahead of sampler / sort).
set -u
TAG=${1:-r02_ak}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python scripts/gpu_probe_stream_prio.py > "$OUT/${TAG}_probe_stream_prio.txt" 2>&1
echo "probe exit $?"; cat "$OUT/${TAG}_probe_stream_prio.txt"
