#!/bin/bash
set -u
TAG=${1:-r01_q}
OUT=$(pwd)/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 90 python -m pytest tests/test_gpu_cf_parity.py -q -k "ndcg or diagnostic" -s > "$OUT/${TAG}_pytest_ndcg.log" 2>&1
echo "pytest ndcg exit $?"; grep "NDCG\|passed\|failed" "$OUT/${TAG}_pytest_ndcg.log" | cut -c1-300
timeout 90 python bench.py --workload als --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_als.json" 2> "$OUT/${TAG}_bench_als.err"
echo "bench als exit $?"; tail -c 1200 "$OUT/${TAG}_bench_als.json"
