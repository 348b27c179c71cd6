#!/usr/bin/env python3
"""GPU probe of the sparse top-k (gorse_sparse_*): all-pairs passes over the IDF vectors of the synthetic datasets under the
library's switches (tile height, split threshold, workgroups per launch, accumulation form); one line per case."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402


def run(name, ptr, idx, val, k=100, tile=0, split=2048, slots=0, atomic=-1, reps=2):
    L = capi.lib()
    L.gorse_hip_test_set_sparse_tile(tile)
    L.gorse_hip_test_set_sparse_split(split)
    L.gorse_hip_test_set_sparse_slots(slots)
    L.gorse_hip_test_set_sparse_atomic(atomic)
    t0 = time.perf_counter()
    s = capi.Sparse(ptr, idx, val)
    t_create = time.perf_counter() - t0
    s.all_pairs(k, fetch=False)
    s.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        s.all_pairs(k, fetch=False)
    wall = (time.perf_counter() - t0) / reps
    n, ms = s.get_profile()
    per = ms / max(n, 1)
    postings, hits = s.last_stats()
    print("%-40s N=%8d nnz=%10d k=%d group=%5s split>%6d slots=%5s atomic=%2d create %7.3f s  all-pairs %9.3f ms (wall %9.3f)  "
          "postings %.3e  %.3e postings/s  %8.1f GB/s algorithmic  non-zero pairs/query %.0f"
          % (name, s.N, int(ptr[-1]), k, tile or "auto", split, slots or "max", atomic, t_create, per, wall * 1e3, postings,
             postings / (per * 1e-3), postings * 8 / (per * 1e-3) / 1e9, hits / s.N), flush=True)
    s.close()
    L.gorse_hip_test_set_sparse_tile(0)
    L.gorse_hip_test_set_sparse_split(2048)
    L.gorse_hip_test_set_sparse_slots(0)
    L.gorse_hip_test_set_sparse_atomic(-1)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    shapes = []
    if which in ("all", "small"):
        shapes.append(("S-ml1m", synth.s_ml1m()))
    if which == "c3tiles":  # the switches that matter at scale, on the C3 shard's item-to-item vectors only
        data = synth.s_big_shard(rank=0, world=8)
        i2i = synth.idf_vectors(data.iptr, data.iidx, data.U)
        run("C3/8 users item-to-item", *i2i)
        for tile in (1024, 4096):
            run("C3/8 users item-to-item", *i2i, tile=tile)
        for split in (512, 8192):
            run("C3/8 users item-to-item", *i2i, split=split)
        run("C3/8 items user-to-user", *synth.idf_vectors(data.uptr, data.uidx, data.I))
        return
    if which in ("all", "c3"):
        shapes.append(("S-big shard (C3/8)", synth.s_big_shard(rank=0, world=8)))
    for name, data in shapes:
        i2i = synth.idf_vectors(data.iptr, data.iidx, data.U)   # item -> users, users' IDF
        u2u = synth.idf_vectors(data.uptr, data.uidx, data.I)   # user -> items, items' IDF
        run(name + " users item-to-item", *i2i)
        run(name + " items user-to-user", *u2u)
        for tile in (1024, 2048, 8192):
            run(name + " users item-to-item", *i2i, tile=tile)
        for split in (256, 1024, 8192):
            run(name + " users item-to-item", *i2i, split=split)
        run(name + " users item-to-item", *i2i, atomic=0)


if __name__ == "__main__":
    main()
