#!/usr/bin/env python3
"""C4 (1M x 128 bf16, cosine, k = 100): the all-pairs pass's stage times and the crc32 of ALL its result rows (indices, distance bits, counts)
-- for an A/B of two builds of the library in one session (scripts/gpu_ab_lib.sh): equal checksums = every row equal.
usage: gpu_probe_topk_c4_crc.py [n_queries]"""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from gorse_amd import capi, synth  # noqa: E402
from gpu_probe_topk_c4 import run  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
N, d, k = 1_000_000, 128, 100
Xb, Xe = synth.s_emb(N, d, 44)
t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
run(t, k, 0, nq, "C4 pass")
run(t, k, 0, nq, "C4 pass")
idx, dist = t.all_pairs(k, 0, nq)
print("crc32 of %d rows: indices %08x distances %08x" % (nq, zlib.crc32(np.ascontiguousarray(idx).tobytes()),
                                                         zlib.crc32(np.ascontiguousarray(dist).view(np.uint32).tobytes())), flush=True)
