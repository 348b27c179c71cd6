#!/bin/bash
# Run on the GPU box (through gpurun): bench line + rocprofv3 kernel stats + separate PMC passes for the
# BPR bench, summaries into gpurun_out/<tag>_*.  Usage: scripts/gpu_profile_bench.sh <tag> [bench args]
set -u
TAG=${1:-r01}
shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py "$@" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
tail -1 "$OUT/${TAG}_bench.json"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline "$@" \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_rocprof.err"
DB=$(find "$OUT/prof_${TAG}" -name '*_results.db' | head -1)
python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_kernel_stats.txt" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_${TAG}_$C" -o bench -- python "$ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline "$@" \
        > /dev/null 2> "$OUT/${TAG}_pmc_$C.err"
    DB=$(find "$OUT/pmc_${TAG}_$C" -name '*_results.db' | head -1)
    python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmc_$C.txt" 2>&1
done
cd "$ROOT"
head -8 "$OUT/${TAG}_kernel_stats.txt"
grep -h "bpr_update" "$OUT/${TAG}_pmc_FETCH_SIZE.txt" "$OUT/${TAG}_pmc_WRITE_SIZE.txt" | tail -4
rm -rf "$OUT"/prof_${TAG} "$OUT"/pmc_${TAG}_*   # databases are large; the summaries are what we keep
