#!/usr/bin/env python3
"""The symmetric sparse pass (C3 shard items) against the library's switches, now that the front delivers: split threshold (= which rows
are the front), heavy threshold (list walk against dense-vector kernel), rows per group, head groups, workgroups.  ms per pass; the first
configuration's rows are the reference every other one is compared with bit for bit."""
import itertools
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402

data = synth.s_big_shard(rank=0, world=8)
ptr, idx, val = synth.idf_vectors(data.iptr, data.iidx, data.U)
L = capi.lib()
N, k = ptr.size - 1, 100
ref = None
lens = np.diff(ptr)


def run(split, heavy, tile, head=-1, slots=0, rows_wgs=0):
    global ref
    L.gorse_hip_test_set_sparse_tile(tile)
    L.gorse_hip_test_set_sparse_split(split)
    L.gorse_hip_test_set_sparse_heavy(heavy)
    L.gorse_hip_test_set_sparse_head(head)
    L.gorse_hip_test_set_sparse_slots(slots)
    L.gorse_hip_test_set_sparse_probe(rows_wgs << 8)
    sp = capi.Sparse(ptr, idx, val)
    out = sp.all_pairs(k, 0, N)
    if ref is None:
        ref = out
    same = all(np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
               for a, b in zip(ref, out))
    t0 = time.perf_counter()
    for _ in range(5):
        sp.all_pairs(k, 0, N, fetch=False)
    sp.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("split %5d (> split: %4d rows) heavy %6d (%3d rows) tile %5d head %2d slots %5d rows-kernel wgs %5d: %7.2f ms  postings %.3e  sym %s  equal %s"
          % (split, int((lens > split).sum()), heavy, int((lens > heavy).sum()), tile or 2048, head, slots or 4096, rows_wgs * 256 or 1024,
             dt * 1e3, sp.last_stats()[0], sp.sym_stats(), same), flush=True)
    sp.close()
    return same


if len(sys.argv) > 3 and sys.argv[1] == "one":  # one configuration (split, heavy), for a kernel timeline
    run(int(sys.argv[2]), int(sys.argv[3]), 0)
    sys.exit(0)
ok = run(2048, 16384, 0)
if len(sys.argv) > 1 and sys.argv[1] == "balance":  # list walk against dense-vector kernel at the default front
    for heavy, wgs in itertools.product((8192, 10240, 12288, 16384, 20480, 24576), (0, 8)):
        ok &= run(2048, heavy, 0, rows_wgs=wgs)
    ok &= run(2048, 16384, 0)
elif len(sys.argv) > 1 and sys.argv[1] == "split":
    for split in (2560, 3072, 3584, 4096, 5120, 6144, 8192, 12288, 16384):
        ok &= run(split, 16384, 0)
    for split, heavy in ((4096, 12288), (4096, 24576), (6144, 12288), (6144, 24576), (3072, 12288), (3072, 24576)):
        ok &= run(split, heavy, 0)
    for split in (3072, 4096, 6144):
        ok &= run(split, 16384, 0, head=0)
    ok &= run(2048, 16384, 0)
else:
    for split, heavy in itertools.product((1024, 1536, 2048, 3072), (8192, 16384, 32768)):
        if (split, heavy) != (2048, 16384):
            ok &= run(split, heavy, 0)
    for tile in (1024, 4096):
        ok &= run(2048, 16384, tile)
        ok &= run(1024 if tile == 1024 else 4096, 16384, tile)
    for head in (0, 2, 4, 8, 16):
        ok &= run(2048, 16384, 0, head=head)
    for slots in (2048, 3072, 8192):
        ok &= run(2048, 16384, 0, slots=slots)
    ok &= run(2048, 16384, 0)
for fn, v in ((L.gorse_hip_test_set_sparse_tile, 0), (L.gorse_hip_test_set_sparse_split, 2048), (L.gorse_hip_test_set_sparse_heavy, 16384),
              (L.gorse_hip_test_set_sparse_head, -1), (L.gorse_hip_test_set_sparse_slots, 0), (L.gorse_hip_test_set_sparse_probe, 0)):
    fn(v)
sys.exit(0 if ok else 1)
