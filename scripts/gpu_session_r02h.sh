#!/bin/bash
# Round 2, eighth device session: wave priorities (ALS solve next to the sibling's fp32 MFMAs, top-k candidate path / epilogue
# next to the sibling's bf16 MFMAs) and the kernel traces r02_g missed.
set -u
TAG=${1:-r02_h}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cf_parity.py tests/test_gpu_baseline_configs.py -q -m gpu -x -k "als or c5" > "$OUT/${TAG}_pytest_als.log" 2>&1
echo "pytest als exit $?"; tail -3 "$OUT/${TAG}_pytest_als.log"
timeout 200 python scripts/gpu_probe_als.py prof > "$OUT/${TAG}_probe_als_prof.txt" 2>&1
echo "probe als prof exit $?"; cut -c1-400 "$OUT/${TAG}_probe_als_prof.txt"
timeout 300 python bench.py --workload als --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_bench_als.json" 2> "$OUT/${TAG}_bench_als.err"
echo "bench als exit $?"; cut -c1-400 "$OUT/${TAG}_bench_als.json"
timeout 400 python scripts/gpu_probe_topk.py prio > "$OUT/${TAG}_probe_topk_prio.txt" 2>&1
echo "probe topk prio exit $?"; cut -c1-330 "$OUT/${TAG}_probe_topk_prio.txt"
cd /tmp
for W in onewarm onecold; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_$W" -o t -- python "$ROOT/scripts/gpu_probe_topk.py" $W > "$OUT/${TAG}_topk_$W.txt" 2> "$OUT/${TAG}_topk_$W.err"
  DB="$(find "$OUT/prof_${TAG}_$W" -name '*_results.db' | head -1)"
  python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_kernel_stats_topk_$W.txt" 2>&1
  echo "== $W"; cut -c1-300 "$OUT/${TAG}_topk_$W.txt"; head -8 "$OUT/${TAG}_kernel_stats_topk_$W.txt" | cut -c1-200
  python - "$DB" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nm = "name" if "name" in cols else cols[0]
for r in cur.execute("select %s, start, end from kernels where %s like '%%sweep%%' order by start" % (nm, nm)):
    print("   sweep launch %.3f ms  %s" % ((r[2] - r[1]) / 1e6, r[0][40:130]))
PY
  rm -rf "$OUT/prof_${TAG}_$W"
done
