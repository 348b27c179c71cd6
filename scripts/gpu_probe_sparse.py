#!/usr/bin/env python3
"""Probe of the sparse top-k (csrc/sparse*) -- `small` leaves the C3 shard out, `tiny` runs S-ml100k only: the "users" item-to-item and "items" user-to-user refresh on three dataset
shapes, host- vs device-built postings, and the number of queries in flight (scratch footprint vs occupancy).
One line per case: create time, all-pairs time by hipEvents, postings/s, algorithmic GB/s (8 B per posting)."""
import sys
import time

import numpy as np
import torch  # noqa: F401  (loads the HIP runtime first)

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402


def run(name, ptr, idx, val, k=100, reps=3, slots=0, device_build=0, heavy=16384, hot=0):
    L = capi.lib()
    L.gorse_hip_test_set_sparse_build(device_build)
    L.gorse_hip_test_set_sparse_slots(slots)
    L.gorse_hip_test_set_sparse_heavy(heavy)
    L.gorse_hip_test_set_sparse_hot(hot)
    t0 = time.perf_counter()
    s = capi.Sparse(ptr, idx, val)
    t_create = time.perf_counter() - t0
    s.all_pairs(k, 0, min(s.N, 4096), fetch=False)
    s.all_pairs(k, fetch=False)
    s.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        s.all_pairs(k, fetch=False)
    wall = (time.perf_counter() - t0) / reps
    n, ms = s.get_profile()
    postings, hits = s.last_stats()
    per = ms / max(n, 1)
    n_heavy = int((np.diff(ptr) > heavy).sum()) if heavy > 0 else 0
    print("%-44s N=%8d nnz=%10d k=%3d slots=%5s heavy>%5d (%5d queries) hot=%4d build=%s create %7.3f s  all-pairs %9.3f ms (wall %9.3f)  postings %.3e  "
          "%.3e postings/s  %8.1f GB/s algorithmic  hit rows/query %.0f"
          % (name, s.N, int(ptr[-1]), k, slots or "max", heavy, n_heavy, hot, "device" if device_build else "host", t_create, per, wall * 1e3,
             postings, postings / (per * 1e-3), postings * 8 / (per * 1e-3) / 1e9, hits / s.N), flush=True)
    s.close()
    L.gorse_hip_test_set_sparse_build(0)
    L.gorse_hip_test_set_sparse_slots(0)
    L.gorse_hip_test_set_sparse_heavy(16384)
    L.gorse_hip_test_set_sparse_hot(0)


def main():
    shapes = [("S-ml100k", synth.s_ml100k())]
    if len(sys.argv) < 2 or sys.argv[1] != "tiny":
        shapes.append(("S-ml1m", synth.s_ml1m()))
    if len(sys.argv) < 2 or sys.argv[1] not in ("small", "tiny"):
        shapes.append(("S-big shard (C3/8)", synth.s_big_shard(rank=0, world=8)))
    for name, data in shapes:
        i2i = synth.idf_vectors(data.iptr, data.iidx, data.U)   # item -> users, users' IDF
        u2u = synth.idf_vectors(data.uptr, data.uidx, data.I)   # user -> items, items' IDF
        run(name + " users item-to-item", *i2i)
        run(name + " users item-to-item", *i2i, device_build=1)
        run(name + " items user-to-user", *u2u)
        for slots in (256, 1024, 4096):
            run(name + " users item-to-item", *i2i, slots=slots)
        for hot in (512, 1024):  # accumulators of the longest rows in LDS
            run(name + " users item-to-item", *i2i, hot=hot)
        for heavy in (0, 2048, 8192, 65536):  # 0 = posting lists only: the longest query sets the launch time
            run(name + " users item-to-item", *i2i, heavy=heavy)


if __name__ == "__main__":
    main()
