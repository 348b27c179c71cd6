#!/bin/bash
# Round 2: four-wave workgroups of the sweep (again: the probe library of r02_v was stale), the scan below 384 queries.
set -u
TAG=${1:-r02_w}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
GORSE_HIP_LIB=$ROOT/gorse_amd/lib/libgorse_hip_w4.so timeout 300 python scripts/gpu_probe_topk.py tiles > "$OUT/${TAG}_probe_topk_w4.txt" 2>&1
echo "w4 exit $?"; cut -c1-260 "$OUT/${TAG}_probe_topk_w4.txt"
GORSE_HIP_LIB=$ROOT/gorse_amd/lib/libgorse_hip_w4.so timeout 400 python -m pytest tests/test_gpu_topk_mfma.py -q -m gpu -x > "$OUT/${TAG}_pytest_w4.log" 2>&1
echo "pytest w4 exit $?"; tail -3 "$OUT/${TAG}_pytest_w4.log"
timeout 400 python -m pytest tests/test_gpu_topk_sgemm.py tests/test_gpu_topk_mfma.py tests/test_gpu_vectors_db.py tests/test_gpu_baseline_configs.py -q -m gpu -x -k "not c3 and not c5 and not c2" > "$OUT/${TAG}_pytest_topk.log" 2>&1
echo "pytest topk exit $?"; tail -3 "$OUT/${TAG}_pytest_topk.log"
timeout 300 python scripts/gpu_probe_query_latency.py > "$OUT/${TAG}_probe_query_latency.txt" 2>&1
echo "latency probe exit $?"; cat "$OUT/${TAG}_probe_query_latency.txt"
