#!/bin/bash
# Round-end GPU session: whole GPU suite, bench lines (default, c3 shard, als), rocprofv3 kernel stats, PMC passes.
# Every step has its own timeout; summaries land in gpurun_out/<tag>_*.
set -u
TAG=${1:-r01_e}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 420 python -m pytest tests -q -m gpu > "$OUT/${TAG}_pytest_gpu.log" 2>&1
echo "pytest gpu exit $?"; tail -3 "$OUT/${TAG}_pytest_gpu.log"
timeout 400 python scripts/gpu_probe_users.py > "$OUT/${TAG}_probe_bpr_users.txt" 2>&1
echo "probe users exit $?"; cat "$OUT/${TAG}_probe_bpr_users.txt"
timeout 300 python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench exit $?"; tail -c 2500 "$OUT/${TAG}_bench.json"; tail -2 "$OUT/${TAG}_bench.err"
python - "$OUT/${TAG}_bench.json" > "$OUT/${TAG}_trafkey.txt" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ml1m_users" if "user runs" in d["config"]["schedule"] else "ml1m")
except Exception:
    print("ml1m")
PY
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_rocprof.err"
DB=$(find "$OUT/prof_${TAG}" -name '*_results.db' | head -1)
python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_kernel_stats.txt" 2>&1
head -14 "$OUT/${TAG}_kernel_stats.txt" | cut -c1-170
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_${TAG}_$C" -o bench -- python "$ROOT/bench.py" --steps 4 --warmup 1 --no-cpu-baseline --no-topk \
        > /dev/null 2> "$OUT/${TAG}_pmc_$C.err"
    DB=$(find "$OUT/pmc_${TAG}_$C" -name '*_results.db' | head -1)
    python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmc_bpr_$C.txt" 2>&1
done
python "$ROOT/scripts/pmc_traffic.py" "$(cat "$OUT/${TAG}_trafkey.txt")" bpr_update "$(find "$OUT/pmc_${TAG}_FETCH_SIZE" -name '*_results.db' | head -1)" \
    "$(find "$OUT/pmc_${TAG}_WRITE_SIZE" -name '*_results.db' | head -1)" "$OUT/${TAG}_traffic.json"
cd "$ROOT"
timeout 240 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_bench_c3.json" 2> "$OUT/${TAG}_bench_c3.err"
echo "bench c3 exit $?"; tail -c 1500 "$OUT/${TAG}_bench_c3.json"
timeout 240 python bench.py --workload als --steps 2 --warmup 1 > "$OUT/${TAG}_bench_als.json" 2> "$OUT/${TAG}_bench_als.err"
echo "bench als exit $?"; tail -c 1800 "$OUT/${TAG}_bench_als.json"; tail -2 "$OUT/${TAG}_bench_als.err"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_als" -o bench -- python "$ROOT/bench.py" --workload als --steps 2 --warmup 1 --no-cpu-baseline \
    > "$OUT/${TAG}_bench_als_under_rocprof.json" 2> "$OUT/${TAG}_rocprof_als.err"
python "$ROOT/scripts/rocpd_summary.py" "$(find "$OUT/prof_${TAG}_als" -name '*_results.db' | head -1)" > "$OUT/${TAG}_kernel_stats_als.txt" 2>&1
head -10 "$OUT/${TAG}_kernel_stats_als.txt" | cut -c1-170
cd "$ROOT"
cd /tmp
# top-k sweep: one SQ pass (MFMA busy, wave cycles, stall buckets, LDS conflicts) and the two HBM passes
timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_${TAG}_topk_SQ" -o bench -- python "$ROOT/bench.py" --workload topk --topk-steps 1 --no-cpu-baseline > /dev/null 2> "$OUT/${TAG}_pmc_topk_SQ.err"
DB=$(find "$OUT/pmc_${TAG}_topk_SQ" -name '*_results.db' | head -1)
python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmc_topk_SQ.txt" 2>&1
grep -h "topk_sweep" "$OUT/${TAG}_pmc_topk_SQ.txt" | cut -c1-60,91-170 | head -12
for C in FETCH_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_${TAG}_topk_$C" -o bench -- python "$ROOT/bench.py" --workload topk --topk-steps 1 --no-cpu-baseline \
        > /dev/null 2> "$OUT/${TAG}_pmc_topk_$C.err"
    DB=$(find "$OUT/pmc_${TAG}_topk_$C" -name '*_results.db' | head -1)
    python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmc_topk_$C.txt" 2>&1
done
cd "$ROOT"
rm -rf "$OUT"/prof_${TAG} "$OUT"/prof_${TAG}_als "$OUT"/pmc_${TAG}_*   # databases are large; the summaries are what we keep
