#!/bin/bash
# Round 2, sixth device session: sparse kernel with the visit pipeline + touched lists, top-k candidate path with lane masks
# (128-row tiles by default), ALS solve without the wave sum.
set -u
TAG=${1:-r02_f}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_vectors_sparse.py -q -m gpu -x > "$OUT/${TAG}_pytest_sparse.log" 2>&1
echo "pytest sparse exit $?"; tail -5 "$OUT/${TAG}_pytest_sparse.log"
timeout 300 python scripts/gpu_probe_sparse_trace.py c3 > "$OUT/${TAG}_probe_sparse_trace.txt" 2>&1
echo "sparse trace exit $?"; cut -c1-300 "$OUT/${TAG}_probe_sparse_trace.txt"
timeout 400 python scripts/gpu_probe_sparse.py c3 > "$OUT/${TAG}_probe_sparse_c3.txt" 2>&1
echo "probe sparse c3 exit $?"; cut -c1-330 "$OUT/${TAG}_probe_sparse_c3.txt"
timeout 600 python -m pytest tests/test_gpu_topk_mfma.py tests/test_gpu_topk_sgemm.py -q -m gpu -x > "$OUT/${TAG}_pytest_topk.log" 2>&1
echo "pytest topk exit $?"; tail -5 "$OUT/${TAG}_pytest_topk.log"
timeout 400 python scripts/gpu_probe_topk.py warm > "$OUT/${TAG}_probe_topk_warm.txt" 2>&1
echo "probe topk warm exit $?"; cut -c1-400 "$OUT/${TAG}_probe_topk_warm.txt"
timeout 300 python scripts/gpu_probe_topk.py prof > "$OUT/${TAG}_probe_topk_prof.txt" 2>&1
echo "probe topk prof exit $?"; cut -c1-600 "$OUT/${TAG}_probe_topk_prof.txt"
timeout 600 python -m pytest tests/test_gpu_cf_parity.py tests/test_gpu_baseline_configs.py -q -m gpu -x -k "als or c5" > "$OUT/${TAG}_pytest_als.log" 2>&1
echo "pytest als exit $?"; tail -5 "$OUT/${TAG}_pytest_als.log"
timeout 300 python bench.py --workload als --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_bench_als.json" 2> "$OUT/${TAG}_bench_als.err"
echo "bench als exit $?"; cut -c1-900 "$OUT/${TAG}_bench_als.json"; tail -2 "$OUT/${TAG}_bench_als.err"
