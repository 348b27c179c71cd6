#!/bin/bash
# Round 2, ninth device session: sparse top-k with two arrangements of the posting lists (groups for ordinary queries,
# stripes for the long ones) and the bisection cut of the ranking buffer; ALS solve sub-phases; top-k without the re-sweep launch.
set -u
TAG=${1:-r02_i}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_vectors_sparse.py -q -m gpu -x > "$OUT/${TAG}_pytest_sparse.log" 2>&1
echo "pytest sparse exit $?"; tail -12 "$OUT/${TAG}_pytest_sparse.log"
timeout 300 python scripts/gpu_probe_sparse_trace.py c3 > "$OUT/${TAG}_probe_sparse_trace.txt" 2>&1
echo "sparse trace exit $?"; cut -c1-300 "$OUT/${TAG}_probe_sparse_trace.txt"
timeout 400 python scripts/gpu_probe_sparse.py c3 > "$OUT/${TAG}_probe_sparse_c3.txt" 2>&1
echo "probe sparse c3 exit $?"; cut -c1-330 "$OUT/${TAG}_probe_sparse_c3.txt"
timeout 200 python scripts/gpu_probe_als.py prof > "$OUT/${TAG}_probe_als_prof.txt" 2>&1
echo "probe als prof exit $?"; cut -c1-460 "$OUT/${TAG}_probe_als_prof.txt"
timeout 600 python -m pytest tests/test_gpu_topk_mfma.py -q -m gpu -x > "$OUT/${TAG}_pytest_topk.log" 2>&1
echo "pytest topk exit $?"; tail -5 "$OUT/${TAG}_pytest_topk.log"
timeout 400 python scripts/gpu_probe_topk.py warm > "$OUT/${TAG}_probe_topk_warm.txt" 2>&1
echo "probe topk warm exit $?"; cut -c1-400 "$OUT/${TAG}_probe_topk_warm.txt"
