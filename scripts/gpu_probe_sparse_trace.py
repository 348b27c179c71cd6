#!/usr/bin/env python3
"""GPU probe: where an all-pairs pass of the sparse top-k spends its time, from the per-work-item records of
gorse_hip_test_sparse_trace: the span of the launch, how busy the resident waves are over it (tail vs throughput), the longest
work items, and the cost per class of work (entries, group parts of long queries, lists-at-once chunks and their rounds, one-list segments)."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    flat = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    data = synth.s_big_shard(rank=0, world=8) if which == "c3" else synth.s_ml1m()
    ptr, idx, val = synth.idf_vectors(data.iptr, data.iidx, data.U)
    capi.lib().gorse_hip_test_set_sparse_tile(tile)
    s = capi.Sparse(ptr, idx, val)
    s.all_pairs(100, fetch=False)
    s.trace(True)
    s.set_profiling(True)
    s.all_pairs(100, fetch=False)
    n, ms = s.get_profile()
    tr = s.trace(False)
    t0, t1 = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64)
    dur = (t1 - t0) / 100.0  # microseconds (100 MHz)
    span = (t1.max() - t0.min()) / 100.0
    print("%s tile=%s flat=%d: launch %.2f ms by hipEvents, %.2f ms first start -> last end; %d work items, sum of their durations %.1f ms "
          "= %.1f waves busy on average" % (which, tile or "auto", flat, ms / max(n, 1), span / 1e3, len(tr), dur.sum() / 1e3, dur.sum() / span))
    # concurrency over time: how many items are running in each 5 % slice of the span
    edges = np.linspace(t0.min(), t1.max(), 21)
    busy = [(np.minimum(t1, edges[i + 1]) - np.maximum(t0, edges[i])).clip(min=0).sum() / (edges[i + 1] - edges[i]) for i in range(20)]
    print("waves busy per 5 %% slice of the span: " + " ".join("%.0f" % b for b in busy))
    order = np.argsort(-dur)[:12]
    print("longest work items (us, query, group+1, entries, chunks at once, rounds, one-list segments, dense groups, re-walked groups, batches, shared rows):")
    for i in order:
        print("  %9.1f %s" % (dur[i], " ".join(str(int(x)) for x in tr[i, 2:])))
    split = tr[:, 3] > 0
    for name, m in (("whole-query items", ~split), ("group parts of long queries", split)):
        if m.any():
            print("%-18s n=%7d  total %.1f ms  entries %.3e  chunks at once %.3e (rounds %.3e)  one-list segments %.3e  dense groups %.3e  re-walked %.3e  batches %.3e  shared rows %.3e"
                  % (name, m.sum(), dur[m].sum() / 1e3, tr[m, 4].sum(), tr[m, 5].sum(), tr[m, 6].sum(), tr[m, 7].sum(), tr[m, 8].sum(), tr[m, 9].sum(),
                     tr[m, 10].sum(), tr[m, 11].sum()))
    for name, m in (("whole-query items", ~split), ("group parts of long queries", split)):
        if m.any():
            print("%-18s ms inside: 64 lists at once %.1f, batches / one list at a time %.1f, read-backs %.1f; until the end of group 8: %.1f"
                  % (name, tr[m, 12].sum() / 1e5, tr[m, 13].sum() / 1e5, tr[m, 14].sum() / 1e5, tr[m, 15].sum() / 1e5))
    whole_e = tr[:, 4]
    for lo, hi in ((0, 64), (64, 256), (256, 1024), (1024, 1 << 30)):
        m = ~split & (whole_e > lo) & (whole_e <= hi)
        if m.any():
            print("whole-query items with %5d < entries <= %-10d n=%7d  total %9.1f ms  mean %8.1f us  of it batches %.1f ms, read-backs %.1f ms, until the end of group 8 %.1f ms"
                  % (lo, hi, m.sum(), dur[m].sum() / 1e3, dur[m].mean(), tr[m, 13].sum() / 1e5, tr[m, 14].sum() / 1e5, tr[m, 15].sum() / 1e5))
    # a linear model of an item's duration in its counters: microseconds per unit
    cols = [np.ones(len(tr)), tr[:, 4], tr[:, 5], tr[:, 6], tr[:, 7], tr[:, 8], tr[:, 9]] + ([tr[:, 10], tr[:, 11]] if flat else [])
    X = np.stack(cols, axis=1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(X, dur, rcond=None)
    print(("least-squares us per: item %.2f, entry %.4f, chunk at once %.3f, round %.3f, one-list segment %.4f, dense group %.2f, re-walked group %.2f"
           + (", batch %.3f, shared row %.3f" if flat else "")) % tuple(coef))

    capi.lib().gorse_hip_test_set_sparse_tile(0)


if __name__ == "__main__":
    main()
