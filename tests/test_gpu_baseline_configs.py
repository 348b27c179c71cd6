"""GPU parity at the BASELINE.json configurations themselves (the sizes bench.py times), through the C ABI:

  C2  S-ml1m 6040 x 3706 x 994,169, nFactors 64, the default Hogwild schedule (user runs): NDCG@10 within +-0.01 of the
      sequential oracle (the bar of model/cf/model_test.go:35-48 is +-0.01 around its anchor), and the top-10 rank lists
      of gorse_mf_rank on the DEVICE's factors equal the oracle's, index for index.
  C3  one 125,000-user shard of S-big (200,000 items, 12.5M feedbacks), nFactors 128: one epoch, factors finite, NDCG@10
      of 8192 held-out users within +-0.01 of the sequential oracle's epoch.
  C3' the whole S-big set (1M x 200K x 100M) on one GPU, nFactors 128: one epoch, NDCG@10 of 8192 held-out users within +-0.01
      of one sequential oracle epoch over the same set (a committed fixture: 100M sequential steps).
  C5  S-als 500,000 x 100,000 x 50M, nFactors 64: one user half-sweep on the device, 2048 rows spread over the row-length
      range (incl. the longest) <= 1e-4 against orc_als_half_range.
      (user half-sweep; round 6: the ITEM half-sweep too -- the long rows' chunk plan and partial reduce at full size).
  C4  S-emb 1,000,000 x 128 bf16, cosine, k = 100: 96 query rows through the scan (path A), 128 rows of a 1024-query call through the
      SQUARE MFMA sweep, 128 rows of a tile-aligned 1024-query call through the SYMMETRIC sweep, and (round 6) THE PASS bench.py
      times -- all_pairs over all 1,000,000 query rows, symmetric form -- with >= 256 rows (first / last query block, tie-replayed rows,
      rows without a pilot threshold, rows whose foreign list overflowed) equal to Bruteforce.SearchIndex restated (oracle) in
      indices AND distance bits.

The element-wise relative error |got - ref| / |ref| is printed next to the bar each ALS comparison uses (`rel_to_scale`:
error over the largest reference magnitude, the form "1e-4 relative fp32" takes for a matrix whose small elements are
differences of large ones)."""
import time

import numpy as np
import pytest

from gorse_amd import capi, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def ndcg(oracle, data, P, Q):
    return float(oracle.evaluate(P, Q, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0])


def test_c2_ml1m_d64_user_runs_ndcg_and_rank_lists(oracle):
    data = synth.s_ml1m()
    d, lr, reg, epochs, seed = 64, 0.05, 0.01, 8, 2024
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 1)
    srt = orc.sort_rows(data.uptr, data.uidx)
    P, Q = P0.copy(), Q0.copy()
    for ep in range(1, epochs + 1):  # sequential (Jobs = 1) epochs on the shared sampler stream
        oracle.bpr_epoch_sampled(P, Q, data.uptr, data.uidx, srt, seed, ep, 0, data.n_train, lr, reg)
    ref = ndcg(oracle, data, P, Q)
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
    mf.set_factors(P0, Q0)
    assert mf.bpr_user_runs()  # the schedule bench.py times at C2
    for ep in range(1, epochs + 1):
        mf.bpr_epoch(data.n_train, lr, reg, seed, ep, mode=capi.BPR_HOGWILD_STORES)
    gP, gQ = mf.get_factors()
    assert np.isfinite(gP).all() and np.isfinite(gQ).all()
    got = ndcg(oracle, data, gP, gQ)
    print("C2 NDCG@10 after %d epochs: sequential oracle %.4f, device user-run schedule %.4f" % (epochs, ref, got))
    assert ref > 0.15
    assert abs(got - ref) < 0.01
    # Rank / TopKFilter (evaluator.go:162-169) on the device's own factors: all 6040 users x 100 candidates
    users, cptr, cand = data.candidates()
    rank, rlen = mf.rank(users, cptr, cand, 10)
    erank, elen = oracle.mf_rank(gP, gQ, users, cptr, cand, 10)
    assert np.array_equal(rlen, elen) and np.array_equal(rank, erank)


def test_c3_shard_d128_one_epoch_ndcg(oracle):
    full = synth.s_big_shard(rank=0, world=8)
    data = synth.hold_out(full, 8192, 99, 5)
    d, lr, reg, seed = 128, 0.05, 0.01, 77
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 1)
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
    mf.set_factors(P0, Q0)
    assert mf.bpr_user_runs()
    mf.bpr_epoch(data.n_train, lr, reg, seed, 1, mode=capi.BPR_HOGWILD_STORES)
    gP, gQ = mf.get_factors()
    assert np.isfinite(gP).all() and np.isfinite(gQ).all()
    got = ndcg(oracle, data, gP, gQ)
    srt = orc.sort_rows(data.uptr, data.uidx)
    P, Q = P0.copy(), Q0.copy()
    t0 = time.perf_counter()
    oracle.bpr_epoch_sampled(P, Q, data.uptr, data.uidx, srt, seed, 1, 0, data.n_train, lr, reg)
    ref = ndcg(oracle, data, P, Q)
    base = ndcg(oracle, data, P0, Q0)
    print("C3 shard NDCG@10 of 8192 held-out users after one epoch: sequential oracle %.4f (%.0f s), device %.4f, untrained %.4f"
          % (ref, time.perf_counter() - t0, got, base))
    assert ref > base + 0.02  # one epoch moved the ranking
    assert abs(got - ref) < 0.01


def test_c3_full_d128_one_epoch_ndcg(oracle):
    """C3 WHOLE on one GPU (1M x 200K x 100M, nFactors 128: P and Q outside the Infinity Cache, 32M-sample chunks): one epoch of
    the default Hogwild schedule, factors finite, NDCG@10 of 8192 held-out users within +-0.01 of ONE SEQUENTIAL ORACLE EPOCH over
    the same set -- 100M sequential SGD steps, minutes of one host core, so the oracle's number is a committed fixture
    (tests/golden/c3full_oracle_ndcg.json, written by scripts/gen_golden_ndcg.py on the CPU from the same seeded
    generators); the evaluation of the device's factors runs here."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "c3full_oracle_ndcg.json")))
    t0 = time.perf_counter()
    data = synth.hold_out(synth.s_big_full(), 8192, 99, 5)
    assert data.n_train == gold["n_train"]  # the same data set the fixture was computed on
    t_data = time.perf_counter() - t0
    d, lr, reg, seed = 128, 0.05, 0.01, 77
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 1)
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
    mf.set_factors(P0, Q0)
    assert mf.bpr_user_runs()
    t0 = time.perf_counter()
    mf.bpr_epoch(data.n_train, lr, reg, seed, 1, mode=capi.BPR_HOGWILD_STORES)
    t_epoch = time.perf_counter() - t0
    gP, gQ = mf.get_factors()
    mf.close()
    assert np.isfinite(gP).all() and np.isfinite(gQ).all()
    got = ndcg(oracle, data, gP, gQ)
    print("C3 whole NDCG@10 of 8192 held-out users after one epoch: sequential oracle %.4f (fixture; %.0f s of one core), device %.4f "
          "(epoch %.3f s incl. first-call allocations), untrained %.4f; data set ready in %.0f s"
          % (gold["ndcg_after_one_epoch"], gold["oracle_epoch_seconds"], got, t_epoch, gold["ndcg_untrained"], t_data))
    assert gold["ndcg_after_one_epoch"] > gold["ndcg_untrained"] + 0.02
    assert abs(got - gold["ndcg_after_one_epoch"]) < 0.01


def test_big_10m_users_d128_one_epoch_ndcg(oracle):
    """north_star's "10M x 1M x 128 synthetic set" (BASELINE.json) on one GPU: 10M users x 1M items, 250M draws = 220M feedbacks
    (the `big` object of bench.py's default line), nFactors 128 -- P is 5.1 GB, Q 512 MB, nothing of the factors fits a cache,
    128M-sample chunks.  One epoch of the default Hogwild schedule: factors finite, NDCG@10 of 8192 held-out users within +-0.01
    of ONE SEQUENTIAL ORACLE EPOCH over the same set (tests/golden/big_oracle_ndcg.json, written by scripts/gen_golden_ndcg.py big
    on the CPU: 2.2e8 sequential steps)."""
    import json
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "big_oracle_ndcg.json")
    if not os.path.exists(path):
        pytest.skip("tests/golden/big_oracle_ndcg.json has not been generated (scripts/gen_golden_ndcg.py big)")
    gold = json.load(open(path))
    t0 = time.perf_counter()
    data = synth.hold_out(synth.s_huge(N=250_000_000), 8192, 99, 5)
    assert data.n_train == gold["n_train"]  # the same data set the fixture was computed on
    t_data = time.perf_counter() - t0
    d, lr, reg, seed = 128, 0.05, 0.01, 77
    P0, Q0 = synth.init_factors_big(data.U, data.I, d, 0.0, 0.001, 1)
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
    mf.set_factors(P0, Q0)
    del P0, Q0
    assert mf.bpr_user_runs()
    t0 = time.perf_counter()
    mf.bpr_epoch(data.n_train, lr, reg, seed, 1, mode=capi.BPR_HOGWILD_STORES)
    t_epoch = time.perf_counter() - t0
    gP, gQ = mf.get_factors()
    mf.close()
    assert np.isfinite(gP).all() and np.isfinite(gQ).all()
    got = ndcg(oracle, data, gP, gQ)
    print("10M x 1M set NDCG@10 of 8192 held-out users after one epoch: sequential oracle %.4f (fixture; %.0f s of one core), device %.4f "
          "(epoch %.3f s incl. first-call allocations), untrained %.4f; data set ready in %.0f s"
          % (gold["ndcg_after_one_epoch"], gold["oracle_epoch_seconds"], got, t_epoch, gold["ndcg_untrained"], t_data))
    assert gold["ndcg_after_one_epoch"] > gold["ndcg_untrained"] + 0.02
    assert abs(got - gold["ndcg_after_one_epoch"]) < 0.01


def rel_to_scale(got, ref):
    return float(np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max())


def elementwise_rel(got, ref):
    ref = ref.astype(np.float64)
    return float((np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)).max())


def _als_rows_fp64(A, B, ptr, idx, bptr, w, reg, rows):
    """the reference's half-sweep recurrence (model.go:659-690 / 707-738) for the given rows in float64, S in float64 too"""
    has = np.diff(bptr) > 0
    B64 = B.astype(np.float64)
    S = B64[has].T @ B64[has]
    out = np.empty((len(rows), A.shape[1]), np.float64)
    for t, u in enumerate(rows):
        Bu = B64[idx[ptr[u]:ptr[u + 1]]]
        pu = A[u].astype(np.float64)
        pred = Bu @ pu
        for f in range(A.shape[1]):
            q = Bu[:, f]
            res = pred - pu[f] * q
            a = ((1 - (1 - w) * res) * q).sum()
            c = ((1 - w) * q * q).sum()
            b = w * (pu @ S[:, f] - pu[f] * S[f, f])
            pu[f] = (a - b) / (c + w * S[f, f] + reg)
            pred = res + pu[f] * q
        out[t] = pu
    return out


def test_c5_als_user_half_sweep_rows(oracle):
    U, I, d, w, reg = 500_000, 100_000, 64, 0.001, 0.06
    uptr, uidx, iptr, iidx = synth.s_als(U, I, 50_000_000, 45)
    P0, Q0 = synth.init_factors(U, I, d, 0.0, 0.1, seed=1)
    mf = capi.MF(U, I, d, uptr, uidx, iptr, iidx)
    mf.set_factors(P0, Q0)
    mf.als_half_epoch(0, w, reg)  # every user row against Q0
    gP, gQ = mf.get_factors()
    assert np.array_equal(gQ.view(np.uint32), Q0.view(np.uint32))
    assert np.isfinite(gP).all()
    lens = np.diff(uptr)
    by_len = np.argsort(lens, kind="stable")
    rows = np.unique(np.concatenate([by_len[np.linspace(0, U - 1, 2040).astype(np.int64)], by_len[-8:]]))  # incl. the longest rows
    A = np.ascontiguousarray(P0[rows])  # the sampled rows as one compact problem: one Gram pass of the oracle for all of them
    sub_ptr = np.zeros(rows.size + 1, np.int64)
    np.cumsum(lens[rows], out=sub_ptr[1:])
    sub_idx = np.concatenate([uidx[uptr[r]:uptr[r + 1]] for r in rows])
    oracle.als_half_range(A, Q0, sub_ptr, sub_idx, iptr, w, reg, 0, rows.size)
    worst_scale = max(rel_to_scale(gP[r:r + 1], A[t:t + 1]) for t, r in enumerate(rows))
    worst_elem = elementwise_rel(gP[rows], A)
    # round 6: the same recurrence in float64 as the arbiter between the two float32 answers (see the item-side test below)
    X = _als_rows_fp64(P0, Q0, uptr, uidx, iptr, w, reg, rows)
    sc = np.abs(X).max(axis=1, keepdims=True)
    e_dev, e_orc = np.abs(gP[rows] - X), np.abs(A - X)
    print("C5 user half-sweep, %d rows (lengths %d..%d): device against oracle: max error / largest |ref| of the row %.2e, element-wise "
          "relative %.2e; against float64 (max error / row scale, element-wise relative): device %.2e, %.2e; oracle %.2e, %.2e"
          % (rows.size, lens[rows].min(), lens[rows].max(), worst_scale, worst_elem, float((e_dev / sc).max()), elementwise_rel(gP[rows], X),
             float((e_orc / sc).max()), elementwise_rel(A, X)))
    assert worst_scale < 1e-4
    # the bar as stated in tests/test_gpu_cf_parity.py: |err| <= 1e-4 |ref| + 5e-5 * (largest |ref| of the row)
    bound = 1e-4 * np.abs(A.astype(np.float64)) + 5e-5 * np.abs(A).max(axis=1, keepdims=True)
    assert (np.abs(gP[rows].astype(np.float64) - A) <= bound).all()
    # and the device is nowhere farther from the exact recurrence than the reference's own float32 arithmetic is (+ 2e-6 of the row scale)
    assert ((e_dev / sc).max(axis=1) <= (e_orc / sc).max(axis=1) + 2e-6).all()


def _check_c4_rows(oracle, Xe, idx, dist, q0, rows, k):
    from concurrent.futures import ThreadPoolExecutor
    import os

    def one(r):
        ei, ed = oracle.search_index(Xe, orc.METRIC_COSINE, q0 + r, k)
        return r, ei, ed
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 1, 32)) as ex:  # the oracle runs outside the GIL (ctypes)
        for r, ei, ed in ex.map(one, rows):
            assert ei.size == k and np.array_equal(idx[r], ei), "row %d: indices differ" % (q0 + r)
            assert np.array_equal(dist[r].view(np.uint32), ed.view(np.uint32)), "row %d: distances differ" % (q0 + r)


def test_c4_rows_against_the_oracle(oracle):
    """96 query rows: a call of fewer than 768 queries takes the literal scan (path A, csrc/topk.hip)."""
    Xb, Xe = synth.s_emb(1_000_000, 128, 44)
    k = 100
    t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
    q0, q1 = 500_000, 500_096
    idx, dist = t.all_pairs(k, q0, q1)
    _check_c4_rows(oracle, Xe, idx, dist, q0, range(q1 - q0), k)


def test_c4_square_sweep_rows_against_the_oracle(oracle):
    """The SQUARE form of the MFMA sweep at C4's size: a call of 1024 queries that does NOT start on a 128-row boundary (500,000 mod
    128 = 32) takes the square sweep (pilot + main sweep, exact rescoring, tie replay: csrc/topk_mfma.hip) -- asserted through the
    handle's profile, which must show sweep launches, and `last_symmetric()`, which must be false -- and 128 of its rows (every eighth)
    equal Bruteforce.SearchIndex restated (common/ann/bruteforce.go:39-83) in indices AND distance bits.  (Until round 5 this was "the
    kernel bench.py times at C4"; since the symmetric form exists, that one is test_c4_full_symmetric_pass_rows_against_the_oracle.)"""
    Xb, Xe = synth.s_emb(1_000_000, 128, 44)
    k = 100
    t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
    q0, q1 = 500_000, 501_024
    t.set_profiling(True)
    idx, dist = t.all_pairs(k, q0, q1)
    launches, sweep_ms = t.get_profile(capi.PROF_TOPK_SWEEP)
    scans, _ = t.get_profile(capi.PROF_TOPK_SCORE)
    n_fb, n_tie = t.last_stats()
    t.set_profiling(False)
    print("C4, 1024 queries from row 500,000: %d sweep launches (%.1f ms), %d scan launches, %d queries through the tie replay, %d fell "
          "back to the scan" % (launches, sweep_ms, scans, n_tie, n_fb))
    assert launches >= 1 and n_fb == 0  # the MFMA sweep answered every query
    assert not t.last_symmetric()
    _check_c4_rows(oracle, Xe, idx, dist, q0, range(0, q1 - q0, 8), k)


def test_c4_symmetric_sweep_of_a_tile_aligned_range_against_the_oracle(oracle):
    """The same 1024-query call started on a 128-row boundary (499,968) takes the SYMMETRIC sweep: 128 rows against the oracle."""
    Xb, Xe = synth.s_emb(1_000_000, 128, 44)
    k = 100
    t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
    q0 = 500_000 // 128 * 128
    q1 = q0 + 1024
    idx, dist = t.all_pairs(k, q0, q1)
    n_fb, n_tie = t.last_stats()
    assert t.last_symmetric() and n_fb == 0
    _check_c4_rows(oracle, Xe, idx, dist, q0, range(0, q1 - q0, 8), k)


def test_c4_full_symmetric_pass_rows_against_the_oracle(oracle):
    """THE pass bench.py times at C4 (its `topk` object): all_pairs(k = 100, 0, 1,000,000) over S-emb 1M x 128 bf16, cosine -- one chunk of
    a million queries through the symmetric sweep (topk_sweep_kernel<8, 2, EP_COARSE, false, 4, false, SYM = true>), the rescoring over own +
    foreign lists, the tie path -- with the results fetched.  Asserted: the symmetric form ran, no query fell back to the scan, and
    >= 256 rows equal Bruteforce.SearchIndex restated (common/ann/bruteforce.go:39-83) in indices AND distance bits.  The sample is drawn
    from what the pass itself reports: rows of the FIRST query block (every neighbour they have in a later block reached them through a
    foreign list) and of the LAST one (no foreign list at all), rows the tie path answered, rows the pilot left without a threshold, rows
    whose foreign list overflowed, and rows spread over the whole range."""
    N, k = 1_000_000, 100
    Xb, Xe = synth.s_emb(N, 128, 44)
    t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
    t0 = time.perf_counter()
    idx, dist = t.all_pairs(k, 0, N)
    t_pass = time.perf_counter() - t0
    n_fb, n_tie = t.last_stats()
    stats = t.sym_stats()
    assert t.last_symmetric()
    assert n_fb == 0  # scan_fallback == 0
    flags = t.last_flags(N)            # non-zero: the query went on to the tie path
    thr = t.warm_thresholds(N)         # +inf: the pilot left the query without a threshold (sym_thresholds_kernel)
    fcnt = t.foreign_counts(N)         # > 512: the foreign list overflowed
    tie_rows = np.flatnonzero(flags)
    unset_rows = np.flatnonzero(np.isposinf(thr))
    over_rows = np.flatnonzero(fcnt > 512)
    print("C4 full symmetric pass (first call, incl. allocations + the fetch of 800 MB): %.2f s; %d queries through the tie path (%d replayed), "
          "%d without a pilot threshold (sym_stats %d), %d foreign lists overflowed (sym_stats %d), %d hits beyond a staging area; "
          "foreign list length mean %.1f max %d" % (t_pass, tie_rows.size, n_tie, unset_rows.size, stats[0], over_rows.size, stats[2],
                                                     stats[3], float(fcnt.mean()), int(fcnt.max())))
    assert tie_rows.size == n_tie  # every flagged query was answered by the replay (none left for the scan)
    # (the rescoring counts an overflowed foreign list only for a query nothing else had flagged before: <=)
    assert unset_rows.size == stats[0] and stats[2] <= over_rows.size
    assert (flags[unset_rows] != 0).all() and (flags[over_rows] != 0).all()  # both kinds take the tie path
    # neighbours in a LATER block exist for the rows of the first block (their answers came through foreign lists)
    first = np.arange(0, 512, 16)
    assert (idx[first] >= 512).any(axis=1).all() and (fcnt[first] > 0).all()
    assert (fcnt[N - 64:] == 0).all()  # nobody sweeps behind the last query block's last tile
    rng = np.random.default_rng(7)
    pick = lambda rows, n: rows if rows.size <= n else rng.choice(rows, n, replace=False)
    sample = np.unique(np.concatenate([
        first, np.arange(N - 512, N, 16),                      # first and last query block
        pick(tie_rows, 64), pick(unset_rows, 32), pick(over_rows, 48),
        np.arange(1000, N, N // 96)]))                         # spread over the range
    assert sample.size >= 256
    _check_c4_rows(oracle, Xe, idx, dist, 0, sample.tolist(), k)
    print("C4 full symmetric pass: %d rows equal the oracle in indices and distance bits (%d of them answered by the tie path, %d without "
          "a pilot threshold, %d with an overflowed foreign list)" % (sample.size, int(np.isin(sample, tie_rows).sum()),
                                                                      int(np.isin(sample, unset_rows).sum()), int(np.isin(sample, over_rows).sum())))


def test_c5_als_item_half_sweep_rows(oracle):
    """C5's ITEM half-sweep at full size (model/cf/model.go:693-738): the side with the 100K+-entry rows, i.e. the chunk plan
    (als_chunk_kernel over 4096-entry chunks), als_partial_reduce_kernel and als_long_solve_kernel at the size bench.py runs them.
    2048 item rows spread over the row-length range incl. the eight longest (4.1M entries), three answers: the device's, the oracle's
    (= the reference's own float32 arithmetic: S summed over 500,000 users one after the other in float32, model.go:695-706, the
    residual recurrence summed over a row's entries one after the other) and the same recurrence in float64.

    What round 6's first run of this test found (profiles/r06_b_probe_gpu_probe_als_c5_items.txt): at this size the REFERENCE's float32
    sums are what drifts -- its answer is 1.9e-4 of the row scale away from float64 already on short rows (S over 500K sequential
    float32 additions), 4.6e-3 on the 4.1M-entry row -- while the device's Gram form (partial sums per 4096-entry chunk, added
    pairwise) stays within 5e-6 of float64 on every row.  "Within 1e-4 of the reference" therefore cannot hold where the reference is
    farther than that from its own exact value; the bars, in order:
      1. device vs float64: the stated ALS bar |err| <= 1e-4 |ref| + 5e-5 rowmax|ref| -- and in fact < 2e-5 of the row scale;
      2. the device is nowhere farther from float64 than the oracle is (+ 2e-6);
      3. device vs oracle: within the stated bar PLUS the oracle's own distance from float64, row by row (triangle inequality)."""
    U, I, d, w, reg = 500_000, 100_000, 64, 0.001, 0.06
    uptr, uidx, iptr, iidx = synth.s_als(U, I, 50_000_000, 45)
    P0, Q0 = synth.init_factors(U, I, d, 0.0, 0.1, seed=1)
    mf = capi.MF(U, I, d, uptr, uidx, iptr, iidx)
    mf.set_factors(P0, Q0)
    mf.als_half_epoch(1, w, reg)  # every item row against P0
    gP, gQ = mf.get_factors()
    assert np.array_equal(gP.view(np.uint32), P0.view(np.uint32))
    assert np.isfinite(gQ).all()
    lens = np.diff(iptr)
    by_len = np.argsort(lens, kind="stable")
    rows = np.unique(np.concatenate([by_len[np.linspace(0, I - 1, 2040).astype(np.int64)], by_len[-8:]]))
    A = np.ascontiguousarray(Q0[rows])  # the sampled rows as one compact problem: one Gram pass of the oracle for all of them
    sub_ptr = np.zeros(rows.size + 1, np.int64)
    np.cumsum(lens[rows], out=sub_ptr[1:])
    sub_idx = np.concatenate([iidx[iptr[r]:iptr[r + 1]] for r in rows])
    # the oracle's half-sweep solves rows of its first argument against the second with the CSR given: items against P0
    oracle.als_half_range(A, P0, sub_ptr, sub_idx, uptr, w, reg, 0, rows.size)
    X = _als_rows_fp64(Q0, P0, iptr, iidx, uptr, w, reg, rows)
    scale = np.abs(X).max(axis=1, keepdims=True)
    e_dev = np.abs(gQ[rows] - X)
    e_orc = np.abs(A - X)
    n_long = int((lens[rows] > 4096).sum())
    print("C5 item half-sweep, %d rows (lengths %d..%d, %d of them cut into chunks): max error / row scale against float64: device %.2e, "
          "oracle (the reference's float32 sums) %.2e; device against oracle %.2e (element-wise relative %.2e)"
          % (rows.size, lens[rows].min(), lens[rows].max(), n_long, float((e_dev / scale).max()), float((e_orc / scale).max()),
             float((np.abs(gQ[rows] - A.astype(np.float64)) / scale).max()), elementwise_rel(gQ[rows], A)))
    assert n_long >= 8 and lens[rows].max() > 100_000  # the chunk plan + partial reduce are on the path
    bar = 1e-4 * np.abs(X) + 5e-5 * scale
    assert (e_dev <= bar).all() and float((e_dev / scale).max()) < 2e-5                      # 1.
    assert ((e_dev / scale).max(axis=1) <= (e_orc / scale).max(axis=1) + 2e-6).all()         # 2.
    assert (np.abs(gQ[rows] - A.astype(np.float64)) <= bar + e_orc.max(axis=1, keepdims=True)).all()  # 3.
