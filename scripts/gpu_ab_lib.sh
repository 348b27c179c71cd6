#!/bin/bash
# A/B of two builds of the library in ONE device session (boxes differ by a few per cent): runs `python scripts/<probe> <args>` alternately with
# GORSE_HIP_LIB = gorse_amd/lib/libgorse_hip.so and every gorse_amd/lib/libgorse_hip_ab_*.so, twice.  usage: gpu_ab_lib.sh <tag> <probe.py> [args ...]
TAG=$1; PROBE=$2; shift 2
OUT=gpurun_out/${TAG}_ab.txt
mkdir -p gpurun_out; : > $OUT
for ROUND in 1 2; do
  for LIB in gorse_amd/lib/libgorse_hip.so gorse_amd/lib/libgorse_hip_ab_*.so; do
    [ -f "$LIB" ] || continue
    echo "== $LIB (round $ROUND)" >> $OUT
    GORSE_HIP_LIB=$PWD/$LIB timeout 600 python scripts/$PROBE "$@" >> $OUT 2>&1
  done
done
cat $OUT
