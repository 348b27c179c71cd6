#!/bin/bash
# Round 2: the sweep's waves decoupled (three tile buffers, counters instead of the tile barrier).
set -u
TAG=${1:-r02_ab}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_topk_mfma.py -q -m gpu -x > "$OUT/${TAG}_pytest_topk.log" 2>&1
echo "pytest topk exit $?"; tail -3 "$OUT/${TAG}_pytest_topk.log"
timeout 200 python scripts/gpu_probe_topk.py tiles > "$OUT/${TAG}_probe_topk_tiles.txt" 2>&1
echo "tiles exit $?"; cut -c1-260 "$OUT/${TAG}_probe_topk_tiles.txt"
timeout 200 python scripts/gpu_probe_topk.py prof > "$OUT/${TAG}_probe_topk_prof.txt" 2>&1
echo "prof exit $?"; cut -c1-420 "$OUT/${TAG}_probe_topk_prof.txt"
timeout 200 python scripts/gpu_probe_topk.py c4 > "$OUT/${TAG}_probe_topk_c4.txt" 2>&1
echo "c4 exit $?"; cut -c1-330 "$OUT/${TAG}_probe_topk_c4.txt"
