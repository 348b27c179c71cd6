#!/usr/bin/env python3
"""The sparse item-to-item pass (C3 shard, 200,000 vectors) in its unsymmetric and its symmetric form (csrc/sparse_kernels.hpp, SymArgs):
ms per pass alternating on one box, postings walked, the symmetric form's statistics, and every row of both compared bit for bit."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402

data = synth.s_big_shard(rank=0, world=8)
ptr, idx, val = synth.idf_vectors(data.iptr, data.iidx, data.U)
sp = capi.Sparse(ptr, idx, val)
L = capi.lib()
N, k = ptr.size - 1, 100
caps = [int(x) for x in sys.argv[1:4]] if len(sys.argv) >= 4 else [0, 0, 0]
res = {}
if len(sys.argv) == 2:  # `<mode> [+ 256 x (workgroups of the heavy-query kernel / 256)]`: that form alone, for a timeline
    L.gorse_hip_test_set_sparse_sym(int(sys.argv[1]) & 0xff, 0, 0, 0)
    L.gorse_hip_test_set_sparse_probe(int(sys.argv[1]) & ~0xff)
    for _ in range(4):
        sp.all_pairs(k, 0, N, fetch=False)
    sp.synchronize()
    print(sp.sym_stats(), sp.last_stats())
    sys.exit(0)
for mode, label in ((0, "unsymmetric"), (1, "symmetric"), (2, "symmetric, no front"), (0, "unsymmetric again"), (1, "symmetric again"),
                    (2, "symmetric, no front again")):
    L.gorse_hip_test_set_sparse_sym(mode, *caps)
    out = sp.all_pairs(k, 0, N)
    res[mode] = out
    t0 = time.perf_counter()
    for _ in range(5):
        sp.all_pairs(k, 0, N, fetch=False)
    sp.synchronize()
    dt = (time.perf_counter() - t0) / 5
    postings, hits = sp.last_stats()
    print("%-20s %7.2f ms per pass, %.3e postings walked, sym stats (ran, rows redone, foreign entries, longest list) %s"
          % (label, dt * 1e3, postings, sp.sym_stats()), flush=True)
L.gorse_hip_test_set_sparse_sym(-1, 0, 0, 0)
same = [bool(np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b))
        for other in (1, 2) for a, b in zip(res[0], res[other])]
print("rows, score bits, counts equal over all %d rows: %s" % (N, same), flush=True)
if not all(same):
    bad = np.nonzero((res[0][0] != res[1][0]).any(axis=1) | (res[0][2] != res[1][2]))[0]
    print("rows that differ: %d, first %s" % (bad.size, bad[:10]))
    for r in bad[:3]:
        print(r, res[0][2][r], res[1][2][r], res[0][0][r][:8], res[1][0][r][:8], res[0][1][r][:4], res[1][1][r][:4])
    sys.exit(1)
