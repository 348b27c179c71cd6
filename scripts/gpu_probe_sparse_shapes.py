#!/usr/bin/env python3
"""The sparse all-pairs pass in its three forms (unsymmetric / symmetric / symmetric without the delivering front) on collections of other
sizes than the C3 shard's items: S-ml100k and S-ml1m items and users, the C3 shard's users.  ms per pass and whether all rows agree."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402

L = capi.lib()
k = 100


def run(label, ptr, idx, val):
    sp = capi.Sparse(ptr, idx, val)
    N = ptr.size - 1
    res, line = {}, []
    for mode in (0, 1, 2, 0, 1, 2):
        L.gorse_hip_test_set_sparse_sym(mode, 0, 0, 0)
        out = sp.all_pairs(k, 0, N)
        res.setdefault(mode, out)
        n = 5 if N > 50000 else 20
        t0 = time.perf_counter()
        for _ in range(n):
            sp.all_pairs(k, 0, N, fetch=False)
        sp.synchronize()
        line.append("%d: %.3f ms" % (mode, (time.perf_counter() - t0) / n * 1e3))
    L.gorse_hip_test_set_sparse_sym(-1, 0, 0, 0)
    same = all(np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
               for other in (1, 2) for a, b in zip(res[0], res[other]))
    print("%-28s %7d rows %9d entries  %s  sym stats %s  all rows equal: %s" % (label, N, int(ptr[-1]), "  ".join(line), sp.sym_stats(), same), flush=True)
    sp.close()
    return same


ok = True
for name, data in (("S-ml100k", synth.s_ml100k()), ("S-ml1m", synth.s_ml1m()), ("C3 shard", synth.s_big_shard(rank=0, world=8))):
    if name != "C3 shard":
        ok &= run(name + " items", *synth.idf_vectors(data.iptr, data.iidx, data.U))
    ok &= run(name + " users", *synth.idf_vectors(data.uptr, data.uidx, data.I))
sys.exit(0 if ok else 1)
