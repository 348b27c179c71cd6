#!/usr/bin/env python3
"""GPU probe: the sparse all-pairs pass (users item-to-item over the C3-shard items) against the rows per group (the LDS accumulators of
a wave: 11/2 bytes per row + the ranking buffer, i.e. how many waves a CU holds).  Results must be identical for every setting.
usage: gpu_probe_sparse_tile.py [rows ...]      Output -> profiles/rNN_*_probe_sparse_tile.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gorse_amd import capi, synth

L = capi.lib()
data = synth.s_big_shard(rank=0, world=8)
ptr, idx, val = synth.idf_vectors(data.iptr, data.iidx, data.U)
N, k = ptr.size - 1, 100
ref = None
for rows in [int(x) for x in sys.argv[1:]] or [2048, 1024, 512, 2048]:
    L.gorse_hip_test_set_sparse_tile(rows)
    sp = capi.Sparse(ptr, idx, val)
    sp.all_pairs(k, 0, min(N, 4096), fetch=False)
    sp.set_profiling(True)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        sp.all_pairs(k, 0, N, fetch=False)
    dt = (time.perf_counter() - t0) / reps
    launches, ms = sp.get_profile()
    postings, hits = sp.last_stats()
    gi, gs, gc = sp.all_pairs(k, 0, N)
    same = ""
    if ref is None:
        ref = (gi, gs, gc)
    else:
        same = "  results identical to the first line: %s" % (np.array_equal(gi, ref[0]) and np.array_equal(gs.view(np.uint32), ref[1].view(np.uint32))
                                                            and np.array_equal(gc, ref[2]))
    print("rows per group %5d: %.2f ms per pass (kernel %.2f ms), %.3e postings = %.1f GB/s algorithmic%s"
          % (rows, dt * 1e3, ms / max(launches, 1), postings, postings * 8 / (ms / max(launches, 1) * 1e-3) / 1e9, same), flush=True)
    sp.close()
L.gorse_hip_test_set_sparse_tile(0)
