#!/bin/bash
# Round 2, fourth device session: sparse top-k with exact clash rounds, the RCCL entry points (per-test timeouts: one of them
# did not finish in r02_c), the warm-started top-k sweep.
set -u
TAG=${1:-r02_d}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_vectors_sparse.py -q -m gpu -x > "$OUT/${TAG}_pytest_sparse.log" 2>&1
echo "pytest sparse exit $?"; tail -5 "$OUT/${TAG}_pytest_sparse.log"
timeout 400 python scripts/gpu_probe_sparse.py c3 > "$OUT/${TAG}_probe_sparse_c3.txt" 2>&1
echo "probe sparse c3 exit $?"; cut -c1-330 "$OUT/${TAG}_probe_sparse_c3.txt"
timeout 600 python -m pytest tests/test_gpu_topk_mfma.py -q -m gpu -x -k "warm or large or all_pairs" > "$OUT/${TAG}_pytest_topk.log" 2>&1
echo "pytest topk exit $?"; tail -8 "$OUT/${TAG}_pytest_topk.log"
timeout 400 python scripts/gpu_probe_topk.py warm > "$OUT/${TAG}_probe_topk_warm.txt" 2>&1
echo "probe topk warm exit $?"; cut -c1-400 "$OUT/${TAG}_probe_topk_warm.txt"
timeout 500 python -m pytest tests/test_gpu_comm.py -v -m gpu --timeout 150 --durations 10 > "$OUT/${TAG}_pytest_comm.log" 2>&1
echo "pytest comm exit $?"; tail -40 "$OUT/${TAG}_pytest_comm.log" | cut -c1-250
