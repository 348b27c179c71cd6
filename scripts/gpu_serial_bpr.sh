#!/bin/bash
# Kernel stats of BPR epochs with the preparation on the update stream (every kernel alone) and in the production schedule.
# usage (through gpurun): scripts/gpu_serial_bpr.sh <tag> <shape> <nFactors> [<nFactors> ...]
set -u
TAG=$1; SHAPE=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for D in "$@"; do
  for MODE in serial overlapped; do
    rocprofv3 --kernel-trace -d $OUT/prof_ser -o tl -- python $ROOT/scripts/gpu_probe_bpr_serial.py $SHAPE $MODE $D > $OUT/${TAG}_serial_${SHAPE}_${D}_${MODE}.txt 2>&1
    DB=$(find $OUT/prof_ser -name '*_results.db' | head -1)
    { echo "== $SHAPE nFactors $D $MODE"; grep "ms per epoch" $OUT/${TAG}_serial_${SHAPE}_${D}_${MODE}.txt; python $ROOT/scripts/rocpd_summary.py $DB | head -14 | cut -c1-60,91-170; } >> $OUT/${TAG}_bpr_serial_${SHAPE}.txt
    rm -rf $OUT/prof_ser $OUT/${TAG}_serial_${SHAPE}_${D}_${MODE}.txt
  done
done
cat $OUT/${TAG}_bpr_serial_${SHAPE}.txt
