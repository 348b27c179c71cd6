#!/usr/bin/env python3
"""The symmetric sparse pass on the C3 shard's USERS (125,000 vectors over 200,000 items, no long rows: every row a whole-query item):
three passes for a kernel timeline, the pass's statistics."""
import sys

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402

data = synth.s_big_shard(rank=0, world=8)
ptr, idx, val = synth.idf_vectors(data.uptr, data.uidx, data.I)
sp = capi.Sparse(ptr, idx, val)
L = capi.lib()
if len(sys.argv) >= 4:
    L.gorse_hip_test_set_sparse_sym(1, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))
for _ in range(3):
    sp.all_pairs(100, 0, ptr.size - 1, fetch=False)
sp.synchronize()
print("sym stats (ran, rows redone, foreign entries, longest list)", sp.sym_stats(), "postings walked, non-zero scores", sp.last_stats())
