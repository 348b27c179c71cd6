#!/bin/bash
# Round 2, fifth device session: where the sparse pass spends its time (per-work-item trace), the top-k sweep with two pilots +
# row votes (phase counters warm vs cold), BPR with the user sort under the update kernel.
set -u
TAG=${1:-r02_e}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python scripts/gpu_probe_sparse_trace.py c3 > "$OUT/${TAG}_probe_sparse_trace.txt" 2>&1
echo "sparse trace exit $?"; cut -c1-300 "$OUT/${TAG}_probe_sparse_trace.txt"
timeout 300 python scripts/gpu_probe_sparse_trace.py c3 1024 > "$OUT/${TAG}_probe_sparse_trace_1024.txt" 2>&1
echo "sparse trace 1024 exit $?"; cut -c1-300 "$OUT/${TAG}_probe_sparse_trace_1024.txt"
timeout 600 python -m pytest tests/test_gpu_topk_mfma.py -q -m gpu -x > "$OUT/${TAG}_pytest_topk.log" 2>&1
echo "pytest topk exit $?"; tail -5 "$OUT/${TAG}_pytest_topk.log"
timeout 400 python scripts/gpu_probe_topk.py warm > "$OUT/${TAG}_probe_topk_warm.txt" 2>&1
echo "probe topk warm exit $?"; cut -c1-400 "$OUT/${TAG}_probe_topk_warm.txt"
timeout 300 python scripts/gpu_probe_topk.py prof > "$OUT/${TAG}_probe_topk_prof.txt" 2>&1
echo "probe topk prof exit $?"; cut -c1-600 "$OUT/${TAG}_probe_topk_prof.txt"
timeout 600 python -m pytest tests/test_gpu_cf_parity.py tests/test_gpu_baseline_configs.py -q -m gpu -x -k "bpr or c2 or c3 or sampler" > "$OUT/${TAG}_pytest_bpr.log" 2>&1
echo "pytest bpr exit $?"; tail -5 "$OUT/${TAG}_pytest_bpr.log"
timeout 300 python scripts/gpu_probe_users.py quick > "$OUT/${TAG}_probe_bpr_users.txt" 2>&1
echo "probe users exit $?"; cut -c1-300 "$OUT/${TAG}_probe_bpr_users.txt"
timeout 300 python scripts/gpu_probe_users.py c3 > "$OUT/${TAG}_probe_bpr_users_c3.txt" 2>&1
echo "probe users c3 exit $?"; cut -c1-300 "$OUT/${TAG}_probe_bpr_users_c3.txt"
