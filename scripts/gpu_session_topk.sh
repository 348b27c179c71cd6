#!/bin/bash
# One gpurun session: top-k path B parity tests, the rest of the GPU suite, top-k probe, bench line, rocprofv3 stats.
# Usage (on the GPU box, through gpurun): scripts/gpu_session_topk.sh <tag>
set -u
TAG=${1:-r01_c}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_topk_mfma.py -x -q > "$OUT/${TAG}_pytest_topk_mfma.log" 2>&1
echo "pytest topk_mfma exit $?"; tail -5 "$OUT/${TAG}_pytest_topk_mfma.log"
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_topk_mfma.py > "$OUT/${TAG}_pytest_gpu.log" 2>&1
echo "pytest gpu exit $?"; tail -5 "$OUT/${TAG}_pytest_gpu.log"
timeout 600 python scripts/gpu_probe_topk.py > "$OUT/${TAG}_probe_topk.txt" 2>&1
echo "probe exit $?"; cat "$OUT/${TAG}_probe_topk.txt" | tail -12
timeout 600 python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench exit $?"; tail -c 3000 "$OUT/${TAG}_bench.json"; tail -3 "$OUT/${TAG}_bench.err"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}" -o bench -- python "$ROOT/bench.py" --workload topk --no-cpu-baseline \
    > "$OUT/${TAG}_bench_topk_under_rocprof.json" 2> "$OUT/${TAG}_rocprof.err"
DB=$(find "$OUT/prof_${TAG}" -name '*_results.db' | head -1)
python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_topk_kernel_stats.txt" 2>&1
head -12 "$OUT/${TAG}_topk_kernel_stats.txt"
rm -rf "$OUT"/prof_${TAG}
cd "$ROOT"
