#!/bin/bash
# Round 2, closing device session: the whole GPU suite (with the tests' printed error figures), the default bench line, its
# rocprofv3 kernel statistics, the PMC passes (HBM traffic of the BPR update at C2 and at the C3 shard; MFMA busy of the sweep),
# the sparse switches, what one query costs.  Every step has its own timeout; summaries land in gpurun_out/<tag>_*.
set -u
TAG=${1:-r02_u}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -s -m gpu > "$OUT/${TAG}_pytest_gpu.log" 2>&1
echo "pytest gpu exit $?"; tail -3 "$OUT/${TAG}_pytest_gpu.log"
timeout 400 python bench.py > "$OUT/${TAG}_bench_default.json" 2> "$OUT/${TAG}_bench_default.err"
echo "bench exit $?"; python - "$OUT/${TAG}_bench_default.json" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print("C2", d["value"], d["ms_per_step"], d["roofline"]["frac"])
        for k in ("topk", "c3", "i2i", "als"):
            if k in d and "value" in d[k]:
                print(k, d[k]["value"], d[k]["ms_per_step"], d[k]["roofline"]["frac"])
            elif k in d:
                print(k, d[k])
PY
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_rocprof.err"
DB=$(find "$OUT/prof_${TAG}" -name '*_results.db' | head -1)
python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_kernel_stats.txt" 2>&1
head -24 "$OUT/${TAG}_kernel_stats.txt" | cut -c1-170
if [ "${PMC_BPR:-1}" = "1" ]; then
for W in ml1m c3; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 150 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_${TAG}_${W}_$C" -o bench -- python "$ROOT/bench.py" --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-topk --no-extra \
        > /dev/null 2> "$OUT/${TAG}_pmc_${W}_$C.err"
    DB=$(find "$OUT/pmc_${TAG}_${W}_$C" -name '*_results.db' | head -1)
    python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmc_${W}_$C.txt" 2>&1
  done
  python "$ROOT/scripts/pmc_traffic.py" "${W}_users" bpr_update "$(find "$OUT/pmc_${TAG}_${W}_FETCH_SIZE" -name '*_results.db' | head -1)" \
      "$(find "$OUT/pmc_${TAG}_${W}_WRITE_SIZE" -name '*_results.db' | head -1)" "$OUT/${TAG}_traffic_${W}.json"
  cat "$OUT/${TAG}_traffic_${W}.json" | head -20
done
fi
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_${TAG}_topk_SQ" -o bench -- python "$ROOT/bench.py" --workload topk --topk-steps 1 --no-cpu-baseline > /dev/null 2> "$OUT/${TAG}_pmc_topk_SQ.err"
DB=$(find "$OUT/pmc_${TAG}_topk_SQ" -name '*_results.db' | head -1)
python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmc_topk_SQ.txt" 2>&1
grep -h "topk_sweep" "$OUT/${TAG}_pmc_topk_SQ.txt" | cut -c1-60,91-170 | head -8
cd "$ROOT"
timeout 300 python scripts/gpu_probe_sparse.py c3tiles > "$OUT/${TAG}_probe_sparse_c3.txt" 2>&1
echo "sparse probe exit $?"; cut -c1-200 "$OUT/${TAG}_probe_sparse_c3.txt"
timeout 300 python scripts/gpu_probe_query_latency.py > "$OUT/${TAG}_probe_query_latency.txt" 2>&1
echo "latency probe exit $?"; cat "$OUT/${TAG}_probe_query_latency.txt"
rm -rf "$OUT"/prof_${TAG} "$OUT"/pmc_${TAG}_*   # databases are large; the summaries are what we keep
