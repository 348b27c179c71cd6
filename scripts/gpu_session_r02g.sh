#!/bin/bash
# Round 2, seventh device session: kernel traces (rocprofv3 --kernel-trace --stats) of the warm / cold top-k sweep and of the
# ALS epoch: which launch costs what.
set -u
TAG=${1:-r02_g}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for W in onewarm onecold; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_$W" -o t -- python "$ROOT/scripts/gpu_probe_topk.py" $W > "$OUT/${TAG}_topk_$W.txt" 2> "$OUT/${TAG}_topk_$W.err"
  python "$ROOT/scripts/rocpd_summary.py" "$(find "$OUT/prof_${TAG}_$W" -name '*_results.db' | head -1)" > "$OUT/${TAG}_kernel_stats_topk_$W.txt" 2>&1
  echo "== $W"; cut -c1-300 "$OUT/${TAG}_topk_$W.txt"; head -12 "$OUT/${TAG}_kernel_stats_topk_$W.txt" | cut -c1-200
  python - "$(find "$OUT/prof_${TAG}_$W" -name '*_results.db' | head -1)" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nm = "name" if "name" in cols else cols[0]
for r in cur.execute("select %s, start, end from kernels where %s like '%%sweep%%' order by start" % (nm, nm)):
    print("   sweep launch %.3f ms  %s" % ((r[2] - r[1]) / 1e6, r[0][:110]))
PY
  rm -rf "$OUT/prof_${TAG}_$W"
done
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_als" -o t -- python "$ROOT/bench.py" --workload als --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_als.json" 2> "$OUT/${TAG}_bench_als.err"
python "$ROOT/scripts/rocpd_summary.py" "$(find "$OUT/prof_${TAG}_als" -name '*_results.db' | head -1)" > "$OUT/${TAG}_kernel_stats_als.txt" 2>&1
echo "== als"; head -12 "$OUT/${TAG}_kernel_stats_als.txt" | cut -c1-200
rm -rf "$OUT/prof_${TAG}_als"
cd "$ROOT"
timeout 200 python scripts/gpu_probe_als.py prof > "$OUT/${TAG}_probe_als_prof.txt" 2>&1
echo "probe als prof exit $?"; cut -c1-400 "$OUT/${TAG}_probe_als_prof.txt"
