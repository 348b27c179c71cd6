#!/usr/bin/env python3
"""Timing probe (results of the probe passes are garbage): the symmetric sparse pass when the rows behind the first row group leave that
group out and the whole-query rows inside it walk everything (gorse_hip_test_set_sparse_probe(2)) -- the list walk of a form in which
the first group delivers its scores to every row -- and the heavy-query kernel with more workgroups (bits 8.. of the probe x 256)."""
import sys
import time

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402

data = synth.s_big_shard(rank=0, world=8)
ptr, idx, val = synth.idf_vectors(data.iptr, data.iidx, data.U)
sp = capi.Sparse(ptr, idx, val)
L = capi.lib()
N, k = ptr.size - 1, 100
for probe, label in ((0, "symmetric pass as shipped"), (2, "first group left out"), (2 | (8 << 8), "... heavy-query kernel 2048 workgroups"),
                     (2 | (16 << 8), "... 4096 workgroups"), (8 << 8, "as shipped, heavy-query kernel 2048 workgroups"), (0, "as shipped again")):
    L.gorse_hip_test_set_sparse_probe(probe)
    sp.all_pairs(k, 0, N, fetch=False)
    sp.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        sp.all_pairs(k, 0, N, fetch=False)
    sp.synchronize()
    dt = (time.perf_counter() - t0) / 5
    postings, hits = sp.last_stats()
    print("%-50s %7.2f ms per pass, %.3e postings walked" % (label, dt * 1e3, postings), flush=True)
L.gorse_hip_test_set_sparse_probe(0)
