#!/usr/bin/env python3
"""GPU probe: the sparse all-pairs pass (users item-to-item over the C3-shard items) against the number of HEAD groups -- the groups a
query of ordinary length visits one by one with directly indexed accumulators; the groups behind them go through hashed
super-visits (csrc/sparse_kernels.hpp).  Every setting must return the same indices and score bits as "no super-visits".
usage: gpu_probe_sparse_head.py [ml1m]      Output -> profiles/rNN_*_probe_sparse_head.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gorse_amd import capi, synth

L = capi.lib()
small = len(sys.argv) > 1 and sys.argv[1] == "ml1m"
data = synth.s_ml1m() if small else synth.s_big_shard(rank=0, world=8)
ptr, idx, val = synth.idf_vectors(data.iptr, data.iidx, data.U)
N, k = ptr.size - 1, 100
ref = None
table = len(sys.argv) > 1 and sys.argv[1] == "table"  # the fill of a super-visit's table instead of the head count
settings = [(1 << 20, 2), (-1, 2), (-1, 3), (-1, 4), (-1, 2)] if table else [(h, 2) for h in ([1 << 20, -1, 0, 1, 2, 4, 8, 16, 32] if not small else [1 << 20, -1, 0, 1])]
for head, cap_shift in settings:
    L.gorse_hip_test_set_sparse_head(head)
    L.gorse_hip_test_set_sparse_table(cap_shift)
    sp = capi.Sparse(ptr, idx, val)
    sp.all_pairs(k, 0, min(N, 4096), fetch=False)
    sp.set_profiling(True)
    t0 = time.perf_counter()
    reps = 3
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
        ok = np.array_equal(gi, ref[0]) and np.array_equal(gs.view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(gc, ref[2])
        same = "  results identical to the first line: %s" % ok
    print("head groups %8d, table takes accumulators >> %d postings: %.2f ms per pass (kernel %.2f ms), %.3e postings = %.1f GB/s algorithmic%s"
          % (head, cap_shift, dt * 1e3, ms / max(launches, 1), postings, postings * 8 / (ms / max(launches, 1) * 1e-3) / 1e9, same), flush=True)
    sp.close()
L.gorse_hip_test_set_sparse_head(-1)
L.gorse_hip_test_set_sparse_table(2)
