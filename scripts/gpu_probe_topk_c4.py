#!/usr/bin/env python3
"""C4 (1M x 128 bf16, cosine, k = 100) on the device: the all-pairs pass with its stage times, a few rows checked against the
oracle, and the switches of the tie path (row slices of the history sweep, the replay's shortcut).  usage: gpu_probe_topk_c4.py [n_queries]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def run(t, k, q0, q1, label, reps=2):
    t.all_pairs(k, q0, q1, fetch=False)
    t.synchronize()
    classes = (capi.PROF_TOPK_SWEEP, capi.PROF_TOPK_SELECT, capi.PROF_TOPK_HIST, capi.PROF_TOPK_REPLAY)
    t.set_profiling(True)
    before = [t.get_profile(c) for c in classes]  # the counters are cumulative
    t0 = time.perf_counter()
    for _ in range(reps):
        t.all_pairs(k, q0, q1, fetch=False)
    t.synchronize()
    dt = (time.perf_counter() - t0) / reps
    stages = [(a[0] - b[0], a[1] - b[1]) for a, b in zip([t.get_profile(c) for c in classes], before)]
    t.set_profiling(False)
    n_fb, n_tie = t.last_stats()
    print("%-34s %8.2f ms per pass | sweep %.2f rescore %.2f history %.2f replay %.2f ms | tie queries %d, to the scan %d"
          % (label, dt * 1e3, *(ms / max(reps, 1) for _, ms in stages), n_tie, n_fb), flush=True)


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1_000_000
    N, d, k = 1_000_000, 128, 100
    Xb, Xe = synth.s_emb(N, d, 44)
    t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
    L = capi.lib()
    run(t, k, 0, nq, "default")
    # parity of a few rows (the GPU tests and bench.py check more)
    o = orc.Oracle()
    o.set_isa(orc.ISA_AVX512)
    idx, dist = t.all_pairs(k, 500_000, 500_016)
    for r in range(16):
        ei, ed = o.search_index(Xe, orc.METRIC_COSINE, 500_000 + r, k)
        assert np.array_equal(idx[r], ei) and np.array_equal(dist[r].view(np.uint32), ed.view(np.uint32)), r
    print("16 rows equal the oracle's", flush=True)
    if "quick" in sys.argv[1:]:
        return
    if "sym" in sys.argv[1:]:  # the symmetric form of the sweep against the square one: stage times, then EVERY row of both passes
        SYM_OFF = 1 << 23
        for rep in range(2):
            for v, label in ((0, "symmetric sweep (default)"), (SYM_OFF, "square sweep (variant bit 23)")):
                L.gorse_hip_test_set_topk_variant(v)
                run(t, k, 0, nq, label, reps=2)
                print("   main sweep symmetric: %s" % t.last_symmetric(), flush=True)
        L.gorse_hip_test_set_topk_variant(0)
        i_sy, d_sy = t.all_pairs(k, 0, nq)
        st = t.last_stats()
        print("symmetric: tie queries %d, to the scan %d, warm starts swept again %d" % (st[1], st[0], t.resweeps()), flush=True)
        L.gorse_hip_test_set_topk_variant(SYM_OFF)
        i_sq, d_sq = t.all_pairs(k, 0, nq)
        st = t.last_stats()
        print("square:    tie queries %d, to the scan %d, warm starts swept again %d" % (st[1], st[0], t.resweeps()), flush=True)
        L.gorse_hip_test_set_topk_variant(0)
        same_i = np.array_equal(i_sy, i_sq)
        same_d = np.array_equal(d_sy.view(np.uint32), d_sq.view(np.uint32))
        print("all %d rows of the two passes equal: indices %s, distance bits %s" % (nq, same_i, same_d), flush=True)
        if not (same_i and same_d):
            bad = np.argwhere((i_sy != i_sq).any(1) | (d_sy.view(np.uint32) != d_sq.view(np.uint32)).any(1)).ravel()
            print("  %d rows differ, first: %s" % (bad.size, bad[:20]), flush=True)
            for q in bad[:3]:
                ei, ed = o.search_index(Xe, orc.METRIC_COSINE, int(q), k)
                print("  row %d: symmetric equals the oracle %s, square equals the oracle %s" % (q, np.array_equal(i_sy[q], ei), np.array_equal(i_sq[q], ei)))
        # a query range inside the index (what a rank of a sharded refresh runs): symmetric only inside its own square
        for (a, b) in ((250_112, 500_096), (0, 131_072)):
            L.gorse_hip_test_set_topk_variant(0)
            i1, d1 = t.all_pairs(k, a, b)
            sym = t.last_symmetric()
            print("   range [%d, %d) symmetric %s: equal to the full pass's rows: %s" % (a, b, sym, np.array_equal(i1, i_sq[a:b]) and
                  np.array_equal(d1.view(np.uint32), d_sq[a:b].view(np.uint32))), flush=True)
            run(t, k, a, b, "range [%d, %d) symmetric" % (a, b), reps=2)
            L.gorse_hip_test_set_topk_variant(SYM_OFF)
            run(t, k, a, b, "range [%d, %d) square" % (a, b), reps=2)
        L.gorse_hip_test_set_topk_variant(0)
        return
    if "symfloor" in sys.argv[1:]:  # what the parts of the symmetric sweep cost (results are garbage: timing only)
        run(t, k, 0, nq, "symmetric sweep", reps=2)
        print("   counters [no pilot threshold, unverified, foreign overflow, staging overflow]: %s" % t.sym_stats(), flush=True)
        for v, label in ((1 << 25, "no tile read along its rows"), (2 << 25, "block + row tests, a hit does nothing"), (3 << 25, "hits staged, never flushed"),
                         (1 << 17, "no block qualifies on the column side (own thresholds +inf)"), ((1 << 17) | (1 << 25), "neither side: the symmetric floor"),
                         ((1 << 23) | (1 << 17), "square sweep, no block qualifies")):
            L.gorse_hip_test_set_topk_variant(v)
            run(t, k, 0, nq, label, reps=2)
        L.gorse_hip_test_set_topk_variant(0)
        return
    if "replay" in sys.argv[1:]:  # the tie replay's counters (they overwrite the sweep's)
        L.gorse_hip_test_set_topk_variant(16)
        run(t, k, 0, nq, "instrumented", reps=1)
        c = t.sweep_profile()
        q = max(c[3], 1)
        print("  replay: %d queries, ticks per query mean %.0f max %d; entries mean %.0f max %d; pushes %.0f, T^gap calls %.0f of which not the identity %.0f, literal T %.0f (max %d) per query; undecided %d"
              % (c[3], c[0] / q, c[1], c[2] / q, c[9], c[4] / q, c[5] / q, c[6] / q, c[7] / q, c[8], c[10]), flush=True)
        L.gorse_hip_test_set_topk_variant(0)
        return
    if "prof" in sys.argv[1:]:  # the instrumented twin of the main sweep: where a wave's cycles go (s_memtime ticks per wave)
        for v, label in ((16, "warm main sweep"), (16 | 32, "warm main sweep, compaction at 224"), (16 | 64, "warm main sweep, compaction at 96"),
                         (16 | 256 | (1 << 17), "cold sweep, no block ever qualifies")):
            L.gorse_hip_test_set_topk_variant(v)
            run(t, k, 0, nq, label + " (instrumented)", reps=1)
            c = t.sweep_profile()
            waves = max(c[7], 1)
            print("  per wave: total %.0f = tile top %.0f + waits %.0f + row blocks %.0f (candidate path %.0f: scale %.0f append %.0f compact %.0f)"
                  % (c[6] / waves, c[0] / waves, c[3] / waves, c[1] / waves, c[2] / waves, c[8] / waves, c[9] / waves, c[10] / waves))
            print("  tile top per wave: buffer wait %.0f, DMA issue %.0f, landing wait %.0f, announcement %.0f" % tuple(x / waves for x in c[12:16]))
            print("  blocks %d, on the candidate path %d (%.2f %%), of those with an append %d (%.1f %%), waves %d; ticks per block %.1f"
                  % (c[4], c[5], 100.0 * c[5] / max(c[4], 1), c[11], 100.0 * c[11] / max(c[5], 1), waves, c[1] / max(c[4], 1)), flush=True)
        L.gorse_hip_test_set_topk_variant(0)
        return
    if "lanes" in sys.argv[1:]:  # the tie replay: queries per wave chosen by the launch's size (default) against always 64 (variant bit 28)
        for v, label in ((0, "queries per replay wave by launch size"), (1 << 28, "64 queries per replay wave"),
                         (0, "by launch size again"), (1 << 28, "64 again")):
            L.gorse_hip_test_set_topk_variant(v)
            run(t, k, 0, nq, label, reps=2)
        L.gorse_hip_test_set_topk_variant(0)
        return
    if "warm" in sys.argv[1:]:  # round 6: the history sweep's slices started from the bounds the main sweep's lists prove (default) against cold (bit 11)
        COLD = 1 << 11
        for rep in range(2):
            for v, label in ((0, "history slices warm (default)"), (COLD, "history slices cold (bit 11)")):
                L.gorse_hip_test_set_topk_variant(v)
                run(t, k, 0, nq, label, reps=2)
        L.gorse_hip_test_set_topk_variant(0)
        i_a, d_a = t.all_pairs(k, 0, nq)
        st_a = t.last_stats()
        L.gorse_hip_test_set_topk_variant(COLD)
        i_b, d_b = t.all_pairs(k, 0, nq)
        st_b = t.last_stats()
        L.gorse_hip_test_set_topk_variant(0)
        same = np.array_equal(i_a, i_b) and np.array_equal(d_a.view(np.uint32), d_b.view(np.uint32))
        print("warm: tie queries %d, to the scan %d; cold: %d, %d" % (st_a[1], st_a[0], st_b[1], st_b[0]), flush=True)
        print("all %d rows with warm-started history slices equal the cold form's: %s" % (nq, same), flush=True)
        assert same
        return
    if "hist" in sys.argv[1:]:  # the tie path's history sweep: workgroups of 64 queries (variant bit 27) against 128, 8 / 1 row slices
        for v, label in ((0, "default (128 queries per history workgroup, 8 slices)"), (1 << 27, "64 queries per history workgroup"),
                         (0, "default again"), (1 << 27, "64 queries again")):
            L.gorse_hip_test_set_topk_variant(v)
            run(t, k, 0, nq, label, reps=2)
        L.gorse_hip_test_set_topk_variant(0)
        return
    if "pilot" in sys.argv[1:]:  # the warm start: stride of the pilot sample, with and without the 1/256 pilot in front of it
        for v, label in ((0, "pilots 1/256 + 1/16 (default)"), (1 << 20, "pilots 1/128 + 1/8"), (1 << 21, "pilots 1/512 + 1/32"),
                         (1 << 22, "pilot 1/16 alone"), ((1 << 21) | (1 << 22), "pilot 1/32 alone"), (256, "no warm start")):
            L.gorse_hip_test_set_topk_variant(v)
            run(t, k, 0, nq, label, reps=2)
        L.gorse_hip_test_set_topk_variant(0)
        return
    if "floor" in sys.argv[1:]:  # what the sweep costs without its candidate path (results are garbage: timing only)
        for v, label in ((256, "cold sweep alone"), (256 | (1 << 17), "cold sweep, no block ever qualifies"),
                         (1 << 17, "main: no block qualifies"), (2 << 17, "main: qualifying blocks do nothing"),
                         (3 << 17, "main: candidates counted, not stored"), (32, "compaction at 224"), (64, "compaction at 96")):
            L.gorse_hip_test_set_topk_variant(v)
            run(t, k, 0, nq, label, reps=1)
        L.gorse_hip_test_set_topk_variant(0)
        return
    for v, label in ((1 << 15, "history sweep in one slice"), (1 << 16, "replay: literal T"), ((1 << 15) | (1 << 16), "one slice + literal T (round 2)"),
                     (256, "no warm start"), (4, "no block-level scale bound")):
        L.gorse_hip_test_set_topk_variant(v)
        run(t, k, 0, nq, label, reps=1)
    L.gorse_hip_test_set_topk_variant(0)


if __name__ == "__main__":
    main()
