#!/bin/bash
# The C++ host twins (gorse_amd/host: model/cf, dataset, heap, vectors.Database, logics, gob) under AddressSanitizer and
# UndefinedBehaviorSanitizer: builds libgorse_host.so instrumented, runs the CPU tests that drive it, restores the library.
# No GPU needed.  Prints the number of sanitizer reports (0 = clean) and pytest's summary.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=$ROOT/gorse_amd/lib/libgorse_host.so
cp "$LIB" /tmp/libgorse_host_orig.so
( cd "$ROOT/gorse_amd/host" && g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer \
    -o "$LIB" gorse_cf.cpp gorse_host_capi.cpp -L../lib -lgorse_hip -Wl,-rpath,"$ROOT/gorse_amd/lib" ) || { cp /tmp/libgorse_host_orig.so "$LIB"; exit 1; }
cd "$ROOT"
LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 \
    timeout 1500 python -m pytest tests/test_host_mirror_cpu.py tests/test_vectors_db_cpu.py tests/test_metrics_cpu.py tests/test_items_blob_cpu.py tests/test_rank_keys_cpu.py tests/test_sparse_row_order_cpu.py \
    -q -s > /tmp/host_sanitizers.log 2>&1
cp /tmp/libgorse_host_orig.so "$LIB"
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' /tmp/host_sanitizers.log)"
tail -1 /tmp/host_sanitizers.log
