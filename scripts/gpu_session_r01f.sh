#!/bin/bash
# short session: whole GPU suite (user-run default, sweep variants), BPR schedule probe, top-k variant probe
set -u
TAG=${1:-r01_f}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 420 python -m pytest tests -q -m gpu > "$OUT/${TAG}_pytest_gpu.log" 2>&1
echo "pytest gpu exit $?"; tail -4 "$OUT/${TAG}_pytest_gpu.log"
timeout 300 python scripts/gpu_probe_users.py > "$OUT/${TAG}_probe_bpr_users.txt" 2>&1
echo "probe users exit $?"; cat "$OUT/${TAG}_probe_bpr_users.txt"
timeout 200 python scripts/gpu_probe_topk.py variants > "$OUT/${TAG}_probe_topk_variants.txt" 2>&1
echo "probe topk exit $?"; cat "$OUT/${TAG}_probe_topk_variants.txt"
