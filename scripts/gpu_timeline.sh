#!/bin/bash
# kernel stats + the last N kernels in time order of one probe under rocprofv3 --kernel-trace.
# usage (through gpurun): scripts/gpu_timeline.sh <tag> <name> <N> <probe.py> [args ...]
set -u
TAG=$1; NAME=$2; N=$3; shift 3
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$ROOT rocprofv3 --kernel-trace -d $OUT/prof_tl_$NAME -o tl -- python $ROOT/scripts/"$@" > $OUT/${TAG}_timeline_${NAME}.txt 2>&1
DB=$(find $OUT/prof_tl_$NAME -name '*_results.db' | head -1)
{ python $ROOT/scripts/rocpd_summary.py $DB | head -24 | cut -c1-60,91-170; python $ROOT/scripts/rocpd_timeline.py $DB $N; } >> $OUT/${TAG}_timeline_${NAME}.txt 2>&1
rm -rf $OUT/prof_tl_$NAME
cut -c1-170 $OUT/${TAG}_timeline_${NAME}.txt
