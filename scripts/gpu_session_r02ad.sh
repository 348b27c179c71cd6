#!/bin/bash
# Round 2: the whole GPU suite after the ALS lane-mask fix (nFactors 32) and the scan's tie queries rerouted to the replay;
# query latencies; the sweep's tile loop without its epilogue (the MFMA loop's own ceiling).
set -u
TAG=${1:-r02_ad}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -s -m gpu > "$OUT/${TAG}_pytest_gpu.log" 2>&1
echo "pytest gpu exit $?"; grep -n "passed\|failed" "$OUT/${TAG}_pytest_gpu.log" | tail -2; grep -n "^FAILED" "$OUT/${TAG}_pytest_gpu.log" | head
timeout 300 python scripts/gpu_probe_query_latency.py > "$OUT/${TAG}_probe_query_latency.txt" 2>&1
echo "latency probe exit $?"; cat "$OUT/${TAG}_probe_query_latency.txt"
timeout 300 python scripts/gpu_probe_topk.py ceiling > "$OUT/${TAG}_probe_topk_ceiling.txt" 2>&1
echo "ceiling exit $?"; cut -c1-330 "$OUT/${TAG}_probe_topk_ceiling.txt"
