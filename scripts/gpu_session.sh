#!/bin/bash
# One device session through gpurun: `scripts/gpu_session.sh <tag> <stage> [<stage> ...]`.  Every stage has its own timeout and
# writes gpurun_out/<tag>_<stage>*; the databases of the profiler are summarised (scripts/rocpd_summary.py,
# scripts/pmc_traffic.py) and deleted, the summaries are what gets copied to profiles/.
# Stages:
#   info                 host cores / memory, GPU memory
#   pytest[:<expr>]      the GPU suite (or `-k <expr>`), output with the tests' printed error figures
#   pyfile:<path>[:<k>]  one test file (optionally -k <k>)
#   bench[:<workload>]   python bench.py [--workload W]            -> <tag>_bench_<W>.json
#   stats[:<workload>]   rocprofv3 --kernel-trace --stats of the same command (no CPU baseline) -> <tag>_kernel_stats_<W>.txt
#   pmc:<workload>:<kernel substring>:<key>[:wide]   FETCH_SIZE and WRITE_SIZE passes -> <tag>_traffic.json[key]
#   sq:<workload>:<kernel substring>                 one SQ pass (MFMA busy, wave cycles, waits) -> <tag>_pmc_SQ_<W>.txt
#   pmcx:<workload>:<kernel substring>:<counters joined by +>[:<name>]   one pass of any counters -> <tag>_pmcx_<W>[_<name>].txt
#   mmpmc / mmsq         floats.MM 4096^3 (scripts/gpu_mm_once.py) under the two traffic passes / one SQ pass
#   probe:<script>[:<args with + for spaces>]        python scripts/<script> args -> <tag>_probe_<script>.txt
#   ranks2[:<workload>]  bench.py as TWO ranks on the one GPU over gloo (functional check of the N > 1 path; numbers meaningless)
#   smoke                __graft_entry__.smoke()
set -u
TAG=$1
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
bench_args() {  # workload name -> bench.py arguments
    case "$1" in
        default) echo "" ;;
        *) echo "--workload $1" ;;
    esac
}
for STAGE in "$@"; do
    IFS=':' read -r KIND A B C D <<< "$STAGE"
    T0=$(date +%s)
    case "$KIND" in
    info)
        { nproc; free -g; rocm-smi --showmeminfo vram 2>/dev/null | head -8; python -c "import numpy; print('numpy', numpy.__version__)"; } > "$OUT/${TAG}_info.txt" 2>&1
        cat "$OUT/${TAG}_info.txt" ;;
    pytest)
        if [ -n "${A:-}" ]; then timeout 900 python -m pytest tests -q -s -m gpu -k "$A" > "$OUT/${TAG}_pytest_gpu.log" 2>&1
        else timeout 1200 python -m pytest tests -q -s -m gpu > "$OUT/${TAG}_pytest_gpu.log" 2>&1; fi
        echo "pytest exit $?"; tail -4 "$OUT/${TAG}_pytest_gpu.log" ;;
    pyfile)
        NAME=$(basename "$A" .py)
        if [ -n "${B:-}" ]; then timeout 900 python -m pytest "$A" -q -s -m gpu -k "$B" > "$OUT/${TAG}_pytest_${NAME}.log" 2>&1
        else timeout 900 python -m pytest "$A" -q -s -m gpu > "$OUT/${TAG}_pytest_${NAME}.log" 2>&1; fi
        echo "pytest $NAME exit $?"; tail -4 "$OUT/${TAG}_pytest_${NAME}.log" ;;
    bench)
        W=${A:-default}
        timeout 1500 python bench.py $(bench_args $W) ${BENCH_EXTRA:-} > "$OUT/${TAG}_bench_${W}.json" 2> "$OUT/${TAG}_bench_${W}.err"
        echo "bench $W exit $?"; python "$ROOT/scripts/bench_summary.py" "$OUT/${TAG}_bench_${W}.json" ;;
    stats)
        W=${A:-default}
        ( cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_${W}" -o bench -- python "$ROOT/bench.py" $(bench_args $W) --no-cpu-baseline ${BENCH_EXTRA:-} \
            > "$OUT/${TAG}_bench_under_rocprof_${W}.json" 2> "$OUT/${TAG}_rocprof_${W}.err" )
        DB=$(find "$OUT/prof_${TAG}_${W}" -name '*_results.db' | head -1)
        python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_kernel_stats_${W}.txt" 2>&1
        head -16 "$OUT/${TAG}_kernel_stats_${W}.txt" | cut -c1-170
        rm -rf "$OUT/prof_${TAG}_${W}" ;;
    pmc)
        W=$A
        for CNT in FETCH_SIZE WRITE_SIZE; do
            ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $CNT -d "$OUT/pmc_${TAG}_${W}_$CNT" -o bench -- python "$ROOT/bench.py" $(bench_args $W) --no-cpu-baseline --no-extra --no-topk --sync-steps 0 ${BENCH_EXTRA:-} \
                > /dev/null 2> "$OUT/${TAG}_pmc_${W}_$CNT.err" )
            DB=$(find "$OUT/pmc_${TAG}_${W}_$CNT" -name '*_results.db' | head -1)
            python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmc_${W}_$CNT.txt" 2>&1
        done
        GORSE_PMC_SESSION="${PMC_SESSION:-$TAG}" python "$ROOT/scripts/pmc_traffic.py" "$C" "$B" "$(find "$OUT/pmc_${TAG}_${W}_FETCH_SIZE" -name '*_results.db' | head -1)" \
            "$(find "$OUT/pmc_${TAG}_${W}_WRITE_SIZE" -name '*_results.db' | head -1)" "$OUT/${TAG}_traffic.json" ${D:-narrow}
        rm -rf "$OUT"/pmc_${TAG}_${W}_* ;;
    sq)
        W=$A
        ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
            -d "$OUT/pmc_${TAG}_${W}_SQ" -o bench -- python "$ROOT/bench.py" $(bench_args $W) --no-cpu-baseline --topk-steps 1 ${BENCH_EXTRA:-} > /dev/null 2> "$OUT/${TAG}_pmc_SQ_${W}.err" )
        DB=$(find "$OUT/pmc_${TAG}_${W}_SQ" -name '*_results.db' | head -1)
        python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmc_SQ_${W}.txt" 2>&1
        grep -h "$B" "$OUT/${TAG}_pmc_SQ_${W}.txt" | cut -c1-60,91-170 | head -12
        rm -rf "$OUT/pmc_${TAG}_${W}_SQ" ;;
    pmcx)  # pmcx:<workload>:<kernel substring>:<counters joined by +>[:<name>]   any counter set in one pass -> <tag>_pmcx_<W>[_name].txt
        W=$A
        CNTS=$(echo "$C" | tr '+' ' ')
        ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $CNTS -d "$OUT/pmcx_${TAG}_${W}" -o bench -- python "$ROOT/bench.py" $(bench_args $W) --no-cpu-baseline --topk-steps 1 ${BENCH_EXTRA:-} > /dev/null 2> "$OUT/${TAG}_pmcx_${W}${D:+_$D}.err" )
        DB=$(find "$OUT/pmcx_${TAG}_${W}" -name '*_results.db' | head -1)
        python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmcx_${W}${D:+_$D}.txt" 2>&1
        grep -h "$B" "$OUT/${TAG}_pmcx_${W}${D:+_$D}.txt" | cut -c1-60,91-200 | head -12
        rm -rf "$OUT/pmcx_${TAG}_${W}" ;;
    mmpmc)  # floats.MM 4096^3 (scripts/gpu_mm_once.py): the FETCH_SIZE / WRITE_SIZE passes -> <tag>_traffic.json[mm]
        for CNT in FETCH_SIZE WRITE_SIZE; do
            ( cd /tmp && PYTHONPATH=$ROOT timeout 600 rocprofv3 --kernel-trace --pmc $CNT -d "$OUT/pmc_${TAG}_mm_$CNT" -o bench -- python "$ROOT/scripts/gpu_mm_once.py" > /dev/null 2> "$OUT/${TAG}_pmc_mm_$CNT.err" )
        done
        GORSE_PMC_SESSION="${PMC_SESSION:-$TAG}" python "$ROOT/scripts/pmc_traffic.py" mm sgemm_mfma "$(find "$OUT/pmc_${TAG}_mm_FETCH_SIZE" -name '*_results.db' | head -1)" \
            "$(find "$OUT/pmc_${TAG}_mm_WRITE_SIZE" -name '*_results.db' | head -1)" "$OUT/${TAG}_traffic.json" wide
        rm -rf "$OUT"/pmc_${TAG}_mm_* ;;
    mmsq)   # the same command under one SQ pass: matrix-pipe busy cycles, wave cycles, waits, the clock (GRBM_GUI_ACTIVE / time)
        ( cd /tmp && PYTHONPATH=$ROOT timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
            -d "$OUT/pmc_${TAG}_mm_SQ" -o bench -- python "$ROOT/scripts/gpu_mm_once.py" > "$OUT/${TAG}_pmc_SQ_mm.out" 2> "$OUT/${TAG}_pmc_SQ_mm.err" )
        DB=$(find "$OUT/pmc_${TAG}_mm_SQ" -name '*_results.db' | head -1)
        python "$ROOT/scripts/rocpd_summary.py" "$DB" > "$OUT/${TAG}_pmc_SQ_mm.txt" 2>&1
        grep -h "sgemm_mfma" "$OUT/${TAG}_pmc_SQ_mm.txt" | cut -c1-60,91-170 | head -12; cat "$OUT/${TAG}_pmc_SQ_mm.out" | tail -3
        rm -rf "$OUT/pmc_${TAG}_mm_SQ" ;;
    probe)
        ARGS=$(echo "${B:-}" | tr '+' ' ')
        PLIB="$ROOT/gorse_amd/lib/libgorse_hip_probe.so"  # the probe build (make -C gorse_amd/csrc probe-lib) when it exists
        [ -f "$PLIB" ] && [ "$PLIB" -nt "$ROOT/gorse_amd/lib/libgorse_hip.so" ] && export GORSE_HIP_LIB="$PLIB"  # never a stale one
        timeout 900 python "scripts/$A" $ARGS > "$OUT/${TAG}_probe_$(basename "$A" .py)${C:+_$C}.txt" 2>&1
        echo "probe $A exit $?"; unset GORSE_HIP_LIB; cut -c1-220 "$OUT/${TAG}_probe_$(basename "$A" .py)${C:+_$C}.txt" | tail -40 ;;
    ranks2)  # FUNCTIONAL check of bench.py's N > 1 path on a one-GPU box: two ranks share the GPU, gloo carries the exchange
        W=${A:-c3}
        GORSE_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
            bench.py --gpus 2 --steps 3 --warmup 1 --topk-n 200000 --topk-steps 1 $(bench_args $W) ${BENCH_EXTRA:-} > "$OUT/${TAG}_bench_ranks2_${W}.json" 2> "$OUT/${TAG}_bench_ranks2_${W}.err"
        echo "ranks2 $W exit $?"; tail -3 "$OUT/${TAG}_bench_ranks2_${W}.err"; python "$ROOT/scripts/bench_summary.py" "$OUT/${TAG}_bench_ranks2_${W}.json" ;;
    smoke)
        timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/${TAG}_smoke.txt" 2>&1
        echo "smoke exit $?"; tail -2 "$OUT/${TAG}_smoke.txt" ;;
    *) echo "unknown stage $STAGE" ;;
    esac
    echo "== $STAGE: $(( $(date +%s) - T0 )) s"
done
