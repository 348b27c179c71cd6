#!/bin/bash
# Round 2: rocprofv3 --kernel-trace --stats of the default bench command on HEAD.
set -u
TAG=${1:-r02_ap}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_rocprof.err"
DB=$(find "$OUT/prof_${TAG}" -name '*_results.db' | head -1)
python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_kernel_stats.txt" 2>&1
head -30 "$OUT/${TAG}_kernel_stats.txt" | cut -c1-170
rm -rf "$OUT"/prof_${TAG}
