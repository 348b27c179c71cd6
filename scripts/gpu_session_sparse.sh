#!/bin/bash
# First device session of the sparse top-k (csrc/sparse*.h*): its GPU tests alone (so a failure there does
# not hide behind the rest of the suite), the i2i bench lines on three shapes, rocprofv3 kernel stats and the two HBM PMC
# passes of sparse_query_kernel.  Every step has its own timeout; summaries land in gpurun_out/<tag>_*.  ~3 GPU-minutes.
set -u
TAG=${1:-r02_a}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
GORSE_GPU_ISOLATED=1 timeout 300 python -m pytest tests/test_gpu_vectors_sparse.py tests/test_gpu_x_model_search.py -q -m gpu > "$OUT/${TAG}_pytest_sparse.log" 2>&1
echo "pytest sparse + model search exit $?"; tail -15 "$OUT/${TAG}_pytest_sparse.log"
for SHAPE in ml100k ml1m c3; do
    timeout 300 python bench.py --workload i2i --i2i-shape $SHAPE --steps 5 --warmup 2 > "$OUT/${TAG}_bench_i2i_$SHAPE.json" 2> "$OUT/${TAG}_bench_i2i_$SHAPE.err"
    echo "bench i2i $SHAPE exit $?"; tail -c 1800 "$OUT/${TAG}_bench_i2i_$SHAPE.json"; tail -2 "$OUT/${TAG}_bench_i2i_$SHAPE.err"
done
timeout 500 python scripts/gpu_probe_sparse.py > "$OUT/${TAG}_probe_sparse.txt" 2>&1
echo "probe sparse exit $?"; cut -c1-230 "$OUT/${TAG}_probe_sparse.txt"
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_i2i" -o bench -- python "$ROOT/bench.py" --workload i2i --steps 5 --warmup 2 --no-cpu-baseline \
    > "$OUT/${TAG}_bench_i2i_under_rocprof.json" 2> "$OUT/${TAG}_rocprof_i2i.err"
python "$ROOT/scripts/rocpd_summary.py" "$(find "$OUT/prof_${TAG}_i2i" -name '*_results.db' | head -1)" > "$OUT/${TAG}_kernel_stats_i2i.txt" 2>&1
head -10 "$OUT/${TAG}_kernel_stats_i2i.txt" | cut -c1-170
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_${TAG}_i2i_$C" -o bench -- python "$ROOT/bench.py" --workload i2i --steps 2 --warmup 1 --no-cpu-baseline \
        > /dev/null 2> "$OUT/${TAG}_pmc_i2i_$C.err"
    python "$ROOT/scripts/rocpd_summary.py" "$(find "$OUT/pmc_${TAG}_i2i_$C" -name '*_results.db' | head -1)" > "$OUT/${TAG}_pmc_i2i_$C.txt" 2>&1
    grep -h "sparse_query" "$OUT/${TAG}_pmc_i2i_$C.txt" | cut -c1-170 | head -4
done
cd "$ROOT"
timeout 400 python scripts/gpu_probe_topk.py variants > "$OUT/${TAG}_probe_topk_variants.txt" 2>&1   # incl. the row-vote variant (bit 7)
echo "probe topk variants exit $?"; tail -5 "$OUT/${TAG}_probe_topk_variants.txt" | cut -c1-220
rm -rf "$OUT"/prof_${TAG}_i2i "$OUT"/pmc_${TAG}_i2i_*   # databases are large; the summaries are what we keep
