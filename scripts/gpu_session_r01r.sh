#!/bin/bash
# closing session of round 1 (budget-bound): bench lines of the final code first, the GPU suite last
set -u
TAG=${1:-r01_r}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 100 python bench.py --cpu-seconds 4 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench exit $?"; tail -c 600 "$OUT/${TAG}_bench.json"
timeout 60 python bench.py --workload als --steps 3 --warmup 1 --cpu-seconds 4 > "$OUT/${TAG}_bench_als.json" 2> "$OUT/${TAG}_bench_als.err"
echo "bench als exit $?"; tail -c 300 "$OUT/${TAG}_bench_als.json"
timeout 70 python -m pytest tests -q -m gpu > "$OUT/${TAG}_pytest_gpu.log" 2>&1
echo "pytest gpu exit $?"; tail -3 "$OUT/${TAG}_pytest_gpu.log"
