#!/bin/bash
# Round 2: the CF tests and smoke() after the last edit of mf.hip (stream creation).
set -u
TAG=${1:-r02_al}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_cf_parity.py tests/test_gpu_comm.py tests/test_gpu_vectors_sparse.py -q -m gpu -x > "$OUT/${TAG}_pytest.log" 2>&1
echo "pytest exit $?"; tail -2 "$OUT/${TAG}_pytest.log"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
