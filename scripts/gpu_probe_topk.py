#!/usr/bin/env python3
"""GPU probe: time top-k path B (MFMA sweep + rescoring) on a few index shapes; prints one line per case."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402


def run(name, X, metric, dtype, k, q0, q1, reps=2):
    t0 = time.perf_counter()
    t = capi.TopK(X, metric, dtype=dtype)
    t_create = time.perf_counter() - t0
    t.all_pairs(k, q0, min(q1, q0 + 4096), fetch=False)
    t.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        t.all_pairs(k, q0, q1, fetch=False)
    dt = (time.perf_counter() - t0) / reps
    ns, sweep = t.get_profile(capi.PROF_TOPK_SWEEP)
    nr, resc = t.get_profile(capi.PROF_TOPK_SELECT)
    na, scan = t.get_profile(capi.PROF_TOPK_SCORE)
    nh, hist = t.get_profile(capi.PROF_TOPK_HIST)
    npl, repl = t.get_profile(capi.PROF_TOPK_REPLAY)
    N, d = X.shape
    pairs = (q1 - q0) * (N - 1)
    kd = d if dtype == capi.DTYPE_BF16 else 3 * d
    print("%-34s N=%8d d=%4d k=%3d nq=%8d wall %8.2f ms (%.3e pairs/s) sweep %8.2f ms (%.1f TFLOP/s on %d-deep operands) "
          "rescore %7.2f ms (%d launches) tie-history-sweep %7.2f ms tie-replay %7.2f ms scan-launches %d scan-fallback %d tie-replayed %d re-swept %d create %.2f s"
          % (name, N, d, k, q1 - q0, dt * 1e3, pairs / dt, sweep / reps, 2.0 * kd * (q1 - q0) * N / (sweep / reps * 1e-3) / 1e12,
             kd, resc / reps, nr, hist / reps, repl / reps, na, t.last_stats()[0], t.last_stats()[1], t.resweeps(), t_create), flush=True)
    t.close()


def main():
    capi.lib().gorse_hip_test_set_topk_path(0)
    Xb, Xe = synth.s_emb(1_000_000, 128, 44)
    if len(sys.argv) > 1 and sys.argv[1] == "c4":  # one bounded case: a quarter of the query rows, then all of them
        run("C4 S-emb bf16 cosine 256K q", Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 262144, reps=1)
        run("C4 S-emb bf16 cosine", Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 1_000_000, reps=1)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "tiles":  # 128- and 64-row tiles (run once per library build: GORSE_HIP_LIB)
        for v, label in [(2, "128-row tiles"), (1, "64-row tiles")]:
            capi.lib().gorse_hip_test_set_topk_variant(v)
            run("C4 256K q: " + label, Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 262144, reps=2)
        capi.lib().gorse_hip_test_set_topk_variant(0)
        return
    if len(sys.argv) > 1 and sys.argv[1] in ("onewarm", "onecold"):  # one configuration, for a kernel trace
        capi.lib().gorse_hip_test_set_topk_variant(0 if sys.argv[1] == "onewarm" else 256)
        run("C4 256K q: " + sys.argv[1], Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 262144, reps=1)
        capi.lib().gorse_hip_test_set_topk_variant(0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "prio":  # wave priority of the candidate path / the epilogue, cold and warm
        for v, label in [(256 | 2048, "cold, default priority"), (256, "cold, candidate path prio 3"), (256 | 8192, "cold, + epilogue prio 1"),
                         (0, "warm, candidate path prio 3"), (8192, "warm, + epilogue prio 1"), (2048, "warm, default priority")]:
            capi.lib().gorse_hip_test_set_topk_variant(v)
            run("C4 256K q: " + label, Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 262144, reps=2)
        capi.lib().gorse_hip_test_set_topk_variant(0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "warm":  # the warm-started sweep against the cold one, with the re-sweep counts
        for v, label in [(256, "cold start"), (0, "warm start"), (2, "warm start, 128-row tiles"), (2 | 256, "cold start, 128-row tiles")]:
            capi.lib().gorse_hip_test_set_topk_variant(v)
            run("C4 256K q: " + label, Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 262144, reps=1)
            run("C4 1M q: " + label, Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 1_000_000, reps=1)
        capi.lib().gorse_hip_test_set_topk_variant(0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "prof":  # phase counters of the instrumented sweep twin (s_memtime ticks)
        for v, label in [(16 | 1, "64-row tiles, warm"), (16 | 1 | 256, "64-row tiles, cold"), (16 | 2, "128-row tiles, warm"),
                         (16 | 2 | 256, "128-row tiles, cold")]:
            capi.lib().gorse_hip_test_set_topk_variant(v)
            t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
            for nq in (131072,):
                t0 = time.perf_counter()
                t.all_pairs(100, 0, nq, fetch=False)
                dt = time.perf_counter() - t0
                c = t.sweep_profile()
                waves = max(c[7], 1)
                tot = c[6] / waves
                print("%-22s nq=%8d wall %8.2f ms | per wave: kernel %.3e ticks = store+prefetch %.1f%% + mfma/epilogue %.1f%% "
                      "(slow paths %.1f%%) + barrier wait %.1f%% | row blocks %d, with a candidate %.1f%%, slow path %.0f ticks each "
                      "= count+exchange %.0f + appends %.0f + compaction %.0f; appending lanes per slow block %.2f"
                      % (label, nq, dt * 1e3, tot, 100.0 * c[0] / c[6], 100.0 * c[1] / c[6], 100.0 * c[2] / c[6],
                         100.0 * c[3] / c[6], c[4], 100.0 * c[5] / max(c[4], 1), c[2] / max(c[5], 1), c[8] / max(c[5], 1),
                         c[9] / max(c[5], 1), c[10] / max(c[5], 1), c[11] / max(c[5], 1)), flush=True)
            t.close()
        capi.lib().gorse_hip_test_set_topk_variant(0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "variants":  # tile height x block-level scale bound, C4 shape
        for v, label in [(1 | 4, "64 rows, exact scale"), (1 | 8, "64 rows, block bound"), (2 | 8, "128 rows, block bound"),
                         (1 | 8 | 32, "64 rows, compact at 2x128"), (1 | 8 | 64, "64 rows, compact at 2x96")]:
            capi.lib().gorse_hip_test_set_topk_variant(v)
            run("C4 256K q: " + label, Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 262144, reps=1)
        capi.lib().gorse_hip_test_set_topk_variant(1 | 8 | 32)
        run("C4 1M q: 64 rows, compact at 2x128", Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 1_000_000, reps=1)
        capi.lib().gorse_hip_test_set_topk_variant(128)
        run("C4 256K q: per-row wave vote in the candidate path", Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 262144, reps=1)
        run("C4 1M q: per-row wave vote in the candidate path", Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 1_000_000, reps=1)
        capi.lib().gorse_hip_test_set_topk_variant(0)
        run("C4 1M q: library default", Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 1_000_000, reps=1)
        return
    run("C4 S-emb bf16 cosine", Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 1_000_000)
    run("C4 S-emb bf16 -dot", Xb, capi.METRIC_NEG_DOT, capi.DTYPE_BF16, 100, 0, 1_000_000)
    run("S-emb bf16 cosine 128K queries", Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 100, 0, 131072)
    run("S-emb bf16 cosine k=10", Xb, capi.METRIC_COSINE, capi.DTYPE_BF16, 10, 0, 262144)
    del Xb, Xe
    rng = np.random.default_rng(3)
    Q = (rng.standard_normal((200_000, 64)) * 0.3).astype(np.float32)
    Q *= rng.lognormal(0, 0.5, (200_000, 1)).astype(np.float32)
    run("item factors fp32 d=64 -dot", Q, capi.METRIC_NEG_DOT, capi.DTYPE_F32, 100, 0, 200_000)
    Q2 = rng.standard_normal((200_000, 128)).astype(np.float32)
    run("fp32 d=128 cosine", Q2, capi.METRIC_COSINE, capi.DTYPE_F32, 100, 0, 200_000)
    Xs = (rng.standard_normal((100_000, 16))).astype(np.float32)
    run("fp32 d=16 -dot", Xs, capi.METRIC_NEG_DOT, capi.DTYPE_F32, 10, 0, 100_000)


if __name__ == "__main__":
    main()
