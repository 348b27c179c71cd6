#!/bin/bash
# Round 2: HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) of the sparse list walk, the ALS row kernel and the top-k sweep;
# the top-k and scan tests on the final library.
set -u
TAG=${1:-r02_ae}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_topk_mfma.py tests/test_gpu_topk_sgemm.py -q -m gpu -x > "$OUT/${TAG}_pytest_topk.log" 2>&1
echo "pytest topk exit $?"; tail -2 "$OUT/${TAG}_pytest_topk.log"
cd /tmp
run_pmc() {  # workload key, kernel substring, bench args...
  W=$1; K=$2; shift 2
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_${TAG}_${W}_$C" -o bench -- python "$ROOT/bench.py" "$@" --no-cpu-baseline > /dev/null 2> "$OUT/${TAG}_pmc_${W}_$C.err"
    DB=$(find "$OUT/pmc_${TAG}_${W}_$C" -name '*_results.db' | head -1)
    python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmc_${W}_$C.txt" 2>&1
  done
  python "$ROOT/scripts/pmc_traffic.py" "$W" "$K" "$(find "$OUT/pmc_${TAG}_${W}_FETCH_SIZE" -name '*_results.db' | head -1)" \
      "$(find "$OUT/pmc_${TAG}_${W}_WRITE_SIZE" -name '*_results.db' | head -1)" "$OUT/${TAG}_traffic_${W}.json" | cut -c1-200
}
run_pmc i2i sparse_tile_kernel --workload i2i --steps 2 --warmup 1
run_pmc als als_row_kernel --workload als --steps 2 --warmup 1
run_pmc topk "topk_sweep_kernel<8, 2, true, false, 4" --workload topk --topk-steps 1
cd "$ROOT"
rm -rf "$OUT"/pmc_${TAG}_*
