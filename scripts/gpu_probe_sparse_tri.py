#!/usr/bin/env python3
"""Timing probe: what a symmetric walk of the WHOLE-QUERY items would leave of the sparse item-to-item pass (C3 shard, 200,000 vectors): every
whole-query item stops at its own row group (gorse_hip_test_set_sparse_probe: results are garbage), long / heavy queries as they are."""
import sys
import time

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402

data = synth.s_big_shard(rank=0, world=8)
ptr, idx, val = synth.idf_vectors(data.iptr, data.iidx, data.U)
sp = capi.Sparse(ptr, idx, val)
L = capi.lib()
N, k = ptr.size - 1, 100
for probe, label in ((0, "the pass as shipped"), (1, "whole-query items stop at their own group"), (0, "as shipped again"), (1, "probe again")):
    L.gorse_hip_test_set_sparse_probe(probe)
    sp.all_pairs(k, 0, N, fetch=False)
    sp.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        sp.all_pairs(k, 0, N, fetch=False)
    sp.synchronize()
    dt = (time.perf_counter() - t0) / 5
    postings, hits = sp.last_stats()
    print("%-44s %7.2f ms per pass, %.3e postings walked" % (label, dt * 1e3, postings), flush=True)
L.gorse_hip_test_set_sparse_probe(0)
