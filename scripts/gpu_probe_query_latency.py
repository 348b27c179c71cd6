#!/usr/bin/env python3
"""GPU probe behind the HNSW decision: what one call of the exact index costs.  (a) the floor of a call
that does next to nothing (64 vectors: launch + copies + synchronisation) = what offloading ONE hop of an HNSW walk (<= 32
neighbour distances) would cost at least; (b) exact top-100 of 1 / 8 / 64 / 512 / 4096 queries against 1,000,000 x 128 bf16
vectors; (c) the CPU oracle's time for the 32 distances of one hop."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def timed(f, reps):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps


def main():
    rng = np.random.default_rng(3)
    small = rng.standard_normal((64, 128)).astype(np.float32)
    t = capi.TopK(small, capi.METRIC_COSINE)
    q = small[:1].copy()
    print("one call on a 64-vector index (the floor of any offloaded hop): %.1f us" % (timed(lambda: t.search_vector(q, 10), 200) * 1e6))
    t.close()
    Xb, Xe = synth.s_emb(1_000_000, 128, 44)
    t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
    for nq in (1, 8, 64, 512, 4096):
        qs = np.ascontiguousarray(Xb[1000:1000 + nq])
        dt = timed(lambda: t.search_vector(qs, 100), 5 if nq < 4096 else 2)
        print("exact top-100 of %4d queries against 1,000,000 x 128 bf16 (host to host): %9.3f ms = %8.1f us per query" % (nq, dt * 1e3, dt * 1e6 / nq))
    t.close()
    o = orc.Oracle()
    hop = np.ascontiguousarray(Xe[:32])
    t0 = time.perf_counter()
    for _ in range(2000):
        o.search_vector(hop, orc.METRIC_COSINE, Xe[77], 32)
    print("CPU oracle, the 32 distances of one HNSW hop + their ranking, one thread: %.2f us" % ((time.perf_counter() - t0) / 2000 * 1e6))


if __name__ == "__main__":
    main()
