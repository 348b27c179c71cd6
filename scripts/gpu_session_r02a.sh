#!/bin/bash
# Round 2, first device session: the un-isolated sparse / model-search GPU modules, where the 12 s of the sparse item-to-item
# pass go (rocprofv3 kernel stats + FETCH/WRITE), and the sweep's variants.  Every step has its own timeout.
set -u
TAG=${1:-r02_a}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_vectors_sparse.py tests/test_gpu_x_model_search.py -q -m gpu > "$OUT/${TAG}_pytest_sparse.log" 2>&1
echo "pytest sparse + model search exit $?"; tail -8 "$OUT/${TAG}_pytest_sparse.log"
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_i2i" -o bench -- python "$ROOT/bench.py" --workload i2i --steps 2 --warmup 1 --no-cpu-baseline \
    > "$OUT/${TAG}_bench_i2i_under_rocprof.json" 2> "$OUT/${TAG}_rocprof_i2i.err"
python "$ROOT/scripts/rocpd_summary.py" "$(find "$OUT/prof_${TAG}_i2i" -name '*_results.db' | head -1)" > "$OUT/${TAG}_kernel_stats_i2i.txt" 2>&1
head -14 "$OUT/${TAG}_kernel_stats_i2i.txt" | cut -c1-190
tail -c 1500 "$OUT/${TAG}_bench_i2i_under_rocprof.json"
cd "$ROOT"
timeout 200 python scripts/gpu_probe_sparse.py small > "$OUT/${TAG}_probe_sparse.txt" 2>&1
echo "probe sparse exit $?"; cut -c1-260 "$OUT/${TAG}_probe_sparse.txt"
timeout 300 python scripts/gpu_probe_topk.py variants > "$OUT/${TAG}_probe_topk_variants.txt" 2>&1
echo "probe topk variants exit $?"; cut -c1-260 "$OUT/${TAG}_probe_topk_variants.txt"
rm -rf "$OUT"/prof_${TAG}_i2i
