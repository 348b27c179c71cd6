#!/bin/bash
# Round 2, second device session: first run of the tiled sparse top-k (LDS accumulators, ds_add_f32) and of the parity tests
# at the BASELINE configurations.  Every step has its own timeout.
set -u
TAG=${1:-r02_b}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_vectors_sparse.py -q -m gpu -x > "$OUT/${TAG}_pytest_sparse.log" 2>&1
echo "pytest sparse exit $?"; tail -25 "$OUT/${TAG}_pytest_sparse.log"
timeout 200 python scripts/gpu_probe_sparse.py small > "$OUT/${TAG}_probe_sparse_small.txt" 2>&1
echo "probe sparse small exit $?"; cut -c1-330 "$OUT/${TAG}_probe_sparse_small.txt"
timeout 400 python scripts/gpu_probe_sparse.py c3 > "$OUT/${TAG}_probe_sparse_c3.txt" 2>&1
echo "probe sparse c3 exit $?"; cut -c1-330 "$OUT/${TAG}_probe_sparse_c3.txt"
timeout 300 python bench.py --workload i2i --steps 3 --warmup 1 > "$OUT/${TAG}_bench_i2i.json" 2> "$OUT/${TAG}_bench_i2i.err"
echo "bench i2i exit $?"; tail -c 2500 "$OUT/${TAG}_bench_i2i.json"; tail -3 "$OUT/${TAG}_bench_i2i.err"
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_i2i" -o bench -- python "$ROOT/bench.py" --workload i2i --steps 3 --warmup 1 --no-cpu-baseline \
    > "$OUT/${TAG}_bench_i2i_under_rocprof.json" 2> "$OUT/${TAG}_rocprof_i2i.err"
python "$ROOT/scripts/rocpd_summary.py" "$(find "$OUT/prof_${TAG}_i2i" -name '*_results.db' | head -1)" > "$OUT/${TAG}_kernel_stats_i2i.txt" 2>&1
head -14 "$OUT/${TAG}_kernel_stats_i2i.txt" | cut -c1-190
rm -rf "$OUT"/prof_${TAG}_i2i
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu -s > "$OUT/${TAG}_pytest_baseline.log" 2>&1
echo "pytest baseline configs exit $?"; tail -30 "$OUT/${TAG}_pytest_baseline.log"
