"""GPU parity of top-k path B (bf16 MFMA candidate sweep + exact rescoring, gorse_amd/csrc/topk_mfma.hip).

The bar is the same as for path A: the indices AND the fp32 distances ann.Bruteforce returns
(common/ann/bruteforce.go:39-83), bit for bit, ties included; the oracle is the checker.  Path B is forced
through the test hook so that small inputs exercise it too.
"""
import numpy as np
import pytest

from gorse_amd import capi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def to_bf16(Xf):
    return (np.ascontiguousarray(Xf, np.float32).view(np.uint32) >> 16).astype(np.uint16)


def from_bf16(Xb):
    return (Xb.astype(np.uint32) << 16).view(np.float32)


@pytest.fixture(autouse=True)
def _paths(oracle):
    oracle.set_isa(orc.ISA_AVX512)
    capi.lib().gorse_hip_test_set_topk_path(2)
    yield
    capi.lib().gorse_hip_test_set_topk_path(0)
    capi.lib().gorse_hip_test_set_topk_variant(0)


@pytest.mark.parametrize("variant", [1, 2, 1 | 4, 1 | 8, 2 | 4, 2 | 8])
@pytest.mark.parametrize("dtype,d,N", [(capi.DTYPE_BF16, 128, 4321), (capi.DTYPE_F32, 40, 3000), (capi.DTYPE_BF16, 64, 130)])
def test_sweep_variants_return_the_same_rows(oracle, variant, dtype, d, N):
    # tile height (64 / 128 candidate rows per barrier) and the block-level row-scale bound of the cosine sweep are
    # performance switches: every combination must reproduce the reference's rows, also with skewed norms (where the
    # library itself would not pick the bound) and with a last tile that is mostly past N
    rng = np.random.default_rng(d + N)
    Xf = rng.standard_normal((N, d)).astype(np.float32)
    Xf *= rng.uniform(0.3, 2.5, (N, 1)).astype(np.float32)
    Xf[::7] *= -1.0
    X = to_bf16(Xf) if dtype == capi.DTYPE_BF16 else Xf
    Xe = from_bf16(X) if dtype == capi.DTYPE_BF16 else Xf
    capi.lib().gorse_hip_test_set_topk_variant(variant)
    k = 25
    for metric in (capi.METRIC_COSINE, capi.METRIC_NEG_DOT, capi.METRIC_EUCLIDEAN):
        t = capi.TopK(X, metric, dtype=dtype)
        idx, dist = t.all_pairs(k)
        qs = np.arange(0, N, 37)
        check_rows(oracle, Xe, metric, qs, k, idx[qs], dist[qs])
        qv = rng.standard_normal((70, d)).astype(np.float32)
        qv = to_bf16(qv) if dtype == capi.DTYPE_BF16 else qv
        qe = from_bf16(qv) if dtype == capi.DTYPE_BF16 else qv
        i2, d2, c2 = t.search_vector(qv, k)
        for r in range(0, 70, 9):
            ei, ed = oracle.search_vector(Xe, metric, qe[r], k)
            assert c2[r] == ei.size and np.array_equal(i2[r, :c2[r]], ei) and np.array_equal(bits(d2[r, :c2[r]]), bits(ed))


def check_rows(oracle, Xe, metric, qs, k, idx, dist, prune0=False):
    for r, q in enumerate(qs):
        ei, ed = oracle.search_index(Xe, metric, int(q), k, prune0)
        n = ei.size
        assert np.array_equal(idx[r, :n], ei), (metric, q)
        assert np.array_equal(bits(dist[r, :n]), bits(ed)), (metric, q)
        assert (idx[r, n:] == -1).all()


@pytest.mark.parametrize("metric", [capi.METRIC_NEG_DOT, capi.METRIC_COSINE, capi.METRIC_EUCLIDEAN])
@pytest.mark.parametrize("dtype", [capi.DTYPE_F32, capi.DTYPE_BF16])
@pytest.mark.parametrize("d,k", [(16, 10), (64, 100), (128, 100), (100, 20), (20, 7), (3, 5)])
def test_all_pairs_matches_oracle(oracle, metric, dtype, d, k):
    rng = np.random.default_rng(1000 * d + 10 * metric + dtype)
    N = 2500
    Xf = rng.standard_normal((N, d)).astype(np.float32)
    Xf *= rng.uniform(0.2, 3.0, (N, 1)).astype(np.float32)  # skewed norms: the dot metric's bound uses the max norm
    if dtype == capi.DTYPE_BF16:
        X = to_bf16(Xf)
        Xe = from_bf16(X)
    else:
        X = Xe = Xf
    t = capi.TopK(X, metric, dtype=dtype)
    idx, dist = t.all_pairs(k)
    n_fb, _ = t.last_stats()
    assert n_fb <= N // 50, "path B handed %d of %d queries to the scan" % (n_fb, N)
    qs = np.arange(0, N, 9)
    check_rows(oracle, Xe, metric, qs, k, idx[qs], dist[qs])
    assert (idx != np.arange(N)[:, None]).all()  # i != q (bruteforce.go:47)
    # and the two paths agree on every row
    capi.lib().gorse_hip_test_set_topk_path(1)
    ia, da = t.all_pairs(k, 0, 600)
    assert np.array_equal(ia, idx[:600]) and np.array_equal(bits(da), bits(dist[:600]))


@pytest.mark.parametrize("path", [2, 3])
def test_ties_inside_the_top_k(oracle, path):
    # small-integer vectors: most queries have equal distances inside their top k+1, where the reference's answer
    # is decided by container/heap mechanics.  path 2: the history sweep + literal heap replay (topk_tie_replay_kernel)
    # decides them; path 3: the replay is switched off and the literal scan over all N vectors (path A) must
    capi.lib().gorse_hip_test_set_topk_path(path)
    rng = np.random.default_rng(4)
    N, d, k = 900, 8, 15
    X = rng.integers(-2, 3, (N, d)).astype(np.float32)
    for metric in (capi.METRIC_NEG_DOT, capi.METRIC_COSINE, capi.METRIC_EUCLIDEAN):
        Xm = X.copy()
        if metric == capi.METRIC_COSINE:
            Xm[(Xm == 0).all(1)] = 1.0  # a zero vector makes the reference's cosine NaN; keep path B eligible
        t = capi.TopK(Xm, metric)
        for prune0 in (False, True):
            qs = np.arange(0, 200)
            idx, dist, cnt = t.search_index(qs, k, prune0)
            for r, q in enumerate(qs):
                ei, ed = oracle.search_index(Xm, metric, int(q), k, prune0)
                assert cnt[r] == ei.size, (metric, prune0, q)
                assert np.array_equal(idx[r, :cnt[r]], ei), (metric, prune0, q)
                assert np.array_equal(bits(dist[r, :cnt[r]]), bits(ed))
        n_scan, n_replay = t.last_stats()
        if path == 2:
            assert n_replay > 0, (n_scan, n_replay)
        else:
            assert n_scan > 0 and n_replay == 0


@pytest.mark.parametrize("metric", [capi.METRIC_NEG_DOT, capi.METRIC_EUCLIDEAN])
@pytest.mark.parametrize("N,d,k,lo,hi", [(6000, 8, 15, -3, 4), (20000, 6, 40, -4, 5), (3000, 4, 100, -2, 3)])
def test_tie_replay_long_history(oracle, N, d, k, lo, hi, metric):
    # many compactions (N >> list capacity) and plenty of equal distances: the replay has to bridge long gaps of
    # unrecorded vectors with T^gap while equal weights sit in the heap (cycle detection), for every query
    rng = np.random.default_rng(N + k)
    X = rng.integers(lo, hi, (N, d)).astype(np.float32)
    t = capi.TopK(X, metric)
    qs = rng.choice(N, 300, replace=False)
    idx, dist, cnt = t.search_index(qs, k)
    n_scan, n_replay = t.last_stats()
    for r, q in enumerate(qs):
        ei, ed = oracle.search_index(X, metric, int(q), k)
        assert cnt[r] == ei.size and np.array_equal(idx[r, :cnt[r]], ei), (q, n_scan, n_replay)
        assert np.array_equal(bits(dist[r, :cnt[r]]), bits(ed))
    assert n_replay > 0
    # search by vector: no self exclusion, same machinery
    qv = X[qs[:60]] + 0.0
    i2, d2, c2 = t.search_vector(qv, k)
    for r in range(60):
        ei, ed = oracle.search_vector(X, metric, qv[r], k)
        assert c2[r] == ei.size and np.array_equal(i2[r, :c2[r]], ei) and np.array_equal(bits(d2[r, :c2[r]]), bits(ed))


V_SLICES8, V_SLICE1, V_REPLAY_LITERAL, V_REPLAY_WAVES, V_INSTRUMENTED = 1 << 14, 1 << 15, 1 << 16, 1 << 19, 16


@pytest.mark.parametrize("variant", [V_SLICES8, V_SLICE1, V_SLICES8 | V_REPLAY_LITERAL, V_SLICE1 | V_REPLAY_LITERAL, V_SLICES8 | V_REPLAY_WAVES,
                                     V_SLICES8 | V_INSTRUMENTED])
@pytest.mark.parametrize("metric", [capi.METRIC_NEG_DOT, capi.METRIC_COSINE, capi.METRIC_EUCLIDEAN])
def test_history_sweep_in_row_slices_and_the_replay_shortcut(oracle, variant, metric):
    """The tie path's two round-3 changes, each against its plain form and the oracle: the history sweep cut into eight row
    slices that start cold and are joined by topk_tie_sort_kernel (thresholds of the earlier slices filter the later ones),
    and the replay's test "T = push +inf, pop leaves this heap as it is" in place of the literal T + snapshot compare; the
    replay with one lane per query (default) against the one with a wave per query (V_REPLAY_WAVES).
    Small-integer vectors: ties everywhere, long gaps of unrecorded rows between the recorded ones."""
    rng = np.random.default_rng(77 + metric)
    N, d, k = 24000, 6, 40
    X = rng.integers(-4, 5, (N, d)).astype(np.float32)
    if metric == capi.METRIC_COSINE:
        X[(X == 0).all(1)] = 1.0
    capi.lib().gorse_hip_test_set_topk_variant(variant)
    t = capi.TopK(X, metric)
    qs = rng.choice(N, 256, replace=False)
    idx, dist, cnt = t.search_index(qs, k)
    n_scan, n_replay = t.last_stats()
    for r, q in enumerate(qs):
        ei, ed = oracle.search_index(X, metric, int(q), k)
        assert cnt[r] == ei.size and np.array_equal(idx[r, :cnt[r]], ei), (q, n_scan, n_replay)
        assert np.array_equal(bits(dist[r, :cnt[r]]), bits(ed))
    assert n_replay > 100, (n_scan, n_replay)  # the replay, not the literal scan, answered them
    if variant & V_INSTRUMENTED:  # the lane-per-query replay and what it passed on to the wave-per-query one
        c = t.sweep_profile()
        print("replay (metric %d): %d queries on lanes, %d T^gap calls, %d not the identity, %d literal T (at most %d in one query)"
              % (metric, c[3], c[5], c[6], c[7], c[8]))
        assert c[3] >= n_replay
    # distinct distances and a few planted ties: the slices' join must keep every row the reference's heap accepted
    Xf = rng.standard_normal((N, 16)).astype(np.float32)
    Xf[5000] = Xf[17]
    Xf[23000] = Xf[17]
    t2 = capi.TopK(Xf, metric)
    i2, d2, c2 = t2.search_index(np.arange(0, 1024), 20)
    for q in (0, 17, 18, 500, 1023):
        ei, ed = oracle.search_index(Xf, metric, q, 20)
        assert c2[q] == ei.size and np.array_equal(i2[q, :c2[q]], ei) and np.array_equal(bits(d2[q, :c2[q]]), bits(ed))


V_COLD_SLICES = 1 << 11


@pytest.mark.parametrize("metric", [capi.METRIC_COSINE, capi.METRIC_NEG_DOT, capi.METRIC_EUCLIDEAN])
def test_history_slices_started_from_the_main_sweeps_lists(oracle, metric):
    """Round 6 (csrc/topk_mfma.hip: tie_warm_kernel, GORSE_HIST_DMA).  A bf16 index of d = 128 (eight k-steps: the history sweep takes the
    main sweep's eight-wave LDS-DMA form) with planted duplicates, eight row slices forced on a small index: every slice but the first
    starts from the bound the main sweep's list entries in front of it prove (default) -- against the slices that start cold (variant
    bit 11), every row of the two forms compared, and a sample of tie rows and plain rows against the oracle."""
    rng = np.random.default_rng(4100 + metric)
    N, d, k, nq = 24000, 128, 30, 2048
    Xf = rng.standard_normal((N, d)).astype(np.float32)
    if metric == capi.METRIC_COSINE:
        Xf /= np.sqrt((Xf * Xf).sum(1))[:, None]
    # duplicates of rows that many queries hold in their top k: ties inside the top k + 1, at the boundary, and late in the index
    for src, dst in ((3, 9000), (3, 23990), (40, 12000), (40, 12001), (700, 23000), (1500, 100), (1999, 15000), (5, 6), (2047, 23999)):
        Xf[dst] = Xf[src]
    Xf[1000:1016] = Xf[1000]  # sixteen equal rows: their own queries tie among themselves
    X = to_bf16(Xf)
    Xe = from_bf16(X)
    out = {}
    for v in (V_SLICES8, V_SLICES8 | V_COLD_SLICES):
        capi.lib().gorse_hip_test_set_topk_variant(v | 512)  # (bit 9: the warm-started main sweep on an index this small)
        t = capi.TopK(X, metric, dtype=capi.DTYPE_BF16)
        idx, dist, cnt = t.search_index(np.arange(nq), k)
        out[v] = (idx.copy(), dist.copy(), cnt.copy(), t.last_stats())
        del t
    capi.lib().gorse_hip_test_set_topk_variant(0)
    (i_w, d_w, c_w, st_w), (i_c, d_c, c_c, st_c) = out[V_SLICES8], out[V_SLICES8 | V_COLD_SLICES]
    print("metric %d: tie replays warm %d cold %d, to the scan %d / %d" % (metric, st_w[1], st_c[1], st_w[0], st_c[0]))
    assert st_w[1] >= 16 and st_c[1] == st_w[1]  # the replay answered the tie queries in both forms
    assert np.array_equal(c_w, c_c) and np.array_equal(i_w, i_c) and np.array_equal(bits(d_w), bits(d_c))
    for q in (0, 3, 5, 6, 40, 100, 700, 1000, 1007, 1015, 1500, 1999, 2047, 1234, 77):
        ei, ed = oracle.search_index(Xe, metric, q, k)
        assert c_w[q] == ei.size and np.array_equal(i_w[q, :c_w[q]], ei), q
        assert np.array_equal(bits(d_w[q, :c_w[q]]), bits(ed)), q


@pytest.mark.parametrize("k", [130, 250])
def test_tie_replay_with_large_k(oracle, k):
    """k + 1 > 128 heap slots per query: the lane-per-query replay then takes 32 queries per workgroup instead of 64 (its LDS
    columns grow with k); small-integer vectors make nearly every query a tie query."""
    rng = np.random.default_rng(500 + k)
    N, d = 9000, 6
    X = rng.integers(-4, 5, (N, d)).astype(np.float32)
    capi.lib().gorse_hip_test_set_topk_variant(V_SLICES8)
    try:
        t = capi.TopK(X, capi.METRIC_NEG_DOT)
        qs = rng.choice(N, 96, replace=False)
        idx, dist, cnt = t.search_index(qs, k)
        n_scan, n_replay = t.last_stats()
    finally:
        capi.lib().gorse_hip_test_set_topk_variant(0)
    for r, q in enumerate(qs):
        ei, ed = oracle.search_index(X, capi.METRIC_NEG_DOT, int(q), k)
        assert cnt[r] == ei.size and np.array_equal(idx[r, :cnt[r]], ei), (q, n_scan, n_replay)
        assert np.array_equal(bits(dist[r, :cnt[r]]), bits(ed))
    assert n_replay > 50, (n_scan, n_replay)


def test_duplicates_and_overflowing_lists(oracle):
    # 700 copies of one vector: every list holds > kCap - 128 equal scores -> overflow flag -> scan
    rng = np.random.default_rng(5)
    d, k = 32, 10
    X = np.concatenate([np.tile(rng.standard_normal((1, d)), (700, 1)), rng.standard_normal((300, d))]).astype(np.float32)
    t = capi.TopK(X, capi.METRIC_NEG_DOT)
    qs = np.array([0, 1, 699, 700, 850, 999])
    full_i, full_d = t.all_pairs(k)
    check_rows(oracle, X, capi.METRIC_NEG_DOT, qs, k, full_i[qs], full_d[qs])


@pytest.mark.parametrize("dtype", [capi.DTYPE_F32, capi.DTYPE_BF16])
def test_search_index_lists_and_search_vector(oracle, dtype):
    rng = np.random.default_rng(6)
    N, d, k = 1300, 64, 25
    Xf = rng.standard_normal((N, d)).astype(np.float32)
    X = to_bf16(Xf) if dtype == capi.DTYPE_BF16 else Xf
    Xe = from_bf16(X) if dtype == capi.DTYPE_BF16 else Xf
    for metric in (capi.METRIC_NEG_DOT, capi.METRIC_COSINE, capi.METRIC_EUCLIDEAN):
        t = capi.TopK(X, metric, dtype=dtype)
        qs = np.concatenate([rng.integers(0, N, 150), [5, 5, N - 1, 0]])  # unordered, with repeats
        idx, dist, cnt = t.search_index(qs, k)
        assert (cnt == k).all()
        check_rows(oracle, Xe, metric, qs[::7], k, idx[::7], dist[::7])
        idx, dist, cnt = t.search_index(qs, k, True)  # prune0: distance <= 0 dropped after the selection
        for r in range(0, qs.size, 11):
            ei, ed = oracle.search_index(Xe, metric, int(qs[r]), k, True)
            assert cnt[r] == ei.size and np.array_equal(idx[r, :cnt[r]], ei) and np.array_equal(bits(dist[r, :cnt[r]]), bits(ed))
        qf = rng.standard_normal((90, d)).astype(np.float32)
        qv = to_bf16(qf) if dtype == capi.DTYPE_BF16 else qf
        qe = from_bf16(qv) if dtype == capi.DTYPE_BF16 else qf
        idx, dist, cnt = t.search_vector(qv, k)
        for r in range(0, 90, 6):
            ei, ed = oracle.search_vector(Xe, metric, qe[r], k)
            assert cnt[r] == ei.size and np.array_equal(idx[r, :cnt[r]], ei) and np.array_equal(bits(dist[r, :cnt[r]]), bits(ed))


@pytest.mark.parametrize("N", [1, 2, 33, 64, 65, 449, 513])
def test_small_and_ragged_sizes(oracle, N):
    # k > N-1, N around the tile (64 rows), list (512) and compaction (448) boundaries
    rng = np.random.default_rng(N)
    d, k = 16, 40
    X = rng.standard_normal((N, d)).astype(np.float32)
    t = capi.TopK(X, capi.METRIC_NEG_DOT)
    idx, dist = t.all_pairs(k)
    qs = np.arange(N)
    check_rows(oracle, X, capi.METRIC_NEG_DOT, qs, k, idx, dist)
    i2, d2, c2 = t.search_index(qs, k)
    assert (c2 == min(k, N - 1)).all() and np.array_equal(i2, idx)


def test_adversarial_order(oracle):
    # scores increasing along the index: every block passes the running threshold (the slow path all the way)
    N, d, k = 3000, 16, 30
    rng = np.random.default_rng(8)
    base = rng.standard_normal(d).astype(np.float32)
    X = (np.linspace(0.1, 4.0, N)[:, None] * base[None, :] + 0.01 * rng.standard_normal((N, d))).astype(np.float32)
    t = capi.TopK(X, capi.METRIC_NEG_DOT)
    idx, dist = t.all_pairs(k)
    qs = np.arange(0, N, 61)
    check_rows(oracle, X, capi.METRIC_NEG_DOT, qs, k, idx[qs], dist[qs])


def test_large_index_against_the_scan():
    # C4's shape at a size the scan can check: N = 200K x 128 bf16, cosine, k = 100.  Path B on 4096 queries
    # must equal path A (the literal replay) on a sample of them; size-independent properties on all rows.
    rng = np.random.default_rng(44)
    N, d, k = 200_000, 128, 100
    Xf = rng.standard_normal((N, d)).astype(np.float32)
    Xf /= np.linalg.norm(Xf, axis=1, keepdims=True)
    Xb = to_bf16(Xf)
    t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
    q0, q1 = 70_000, 74_096
    idx, dist = t.all_pairs(k, q0, q1)
    assert t.last_stats()[0] <= 8
    assert (idx >= 0).all() and (idx < N).all()
    assert (idx != np.arange(q0, q1)[:, None]).all()
    assert (np.diff(dist, axis=1) >= 0).all()  # ascending distances
    assert all(np.unique(r).size == k for r in idx[::97])
    sample = np.arange(q0, q1, 409)
    capi.lib().gorse_hip_test_set_topk_path(1)
    ia, da, _ = t.search_index(sample, k)
    assert np.array_equal(ia, idx[sample - q0]) and np.array_equal(bits(da), bits(dist[sample - q0]))


WARM_OFF, WARM_ALWAYS, WARM_SABOTAGE = 256, 512, 512 | 1024


@pytest.mark.parametrize("metric", [capi.METRIC_COSINE, capi.METRIC_NEG_DOT, capi.METRIC_EUCLIDEAN])
@pytest.mark.parametrize("dtype,d,N,k", [(capi.DTYPE_BF16, 128, 40000, 100), (capi.DTYPE_F32, 64, 20000, 30)])
def test_warm_started_sweep_returns_the_same_rows(oracle, metric, dtype, d, N, k):
    """The pilot sweep (every 16th row tile, a small kth) only proposes initial thresholds; the main sweep verifies each one
    and the host sweeps the failures again from -inf.  With the warm start forced on (it is automatic from 2^17 rows), with
    it off, and with a pilot that proposes thresholds far too high (kth = 2: most queries must be swept again), every row
    and every distance bit is the same -- and equal to the oracle's on a sample; rows sorted by norm make the systematic
    sample a biased one."""
    rng = np.random.default_rng(N + d)
    Xf = rng.standard_normal((N, d)).astype(np.float32)
    Xf *= np.sort(rng.uniform(0.5, 2.0, N)).astype(np.float32)[:, None]
    Xf[100:140] = Xf[5]  # duplicates: ties in the top k of their queries
    X = to_bf16(Xf) if dtype == capi.DTYPE_BF16 else Xf
    Xe = from_bf16(X) if dtype == capi.DTYPE_BF16 else Xf
    t = capi.TopK(X, metric, dtype=dtype)
    out = {}
    for v in (WARM_OFF, WARM_ALWAYS, WARM_SABOTAGE):
        capi.lib().gorse_hip_test_set_topk_variant(v)
        out[v] = t.all_pairs(k) + (t.resweeps(),)
    assert out[WARM_OFF][2] == 0 and out[WARM_ALWAYS][2] < N // 10 and out[WARM_SABOTAGE][2] > N // 4
    for v in (WARM_ALWAYS, WARM_SABOTAGE):
        assert np.array_equal(out[v][0], out[WARM_OFF][0]) and np.array_equal(bits(out[v][1]), bits(out[WARM_OFF][1]))
    qs = np.concatenate([np.arange(0, N, 1999), [5, 100, 139]])
    check_rows(oracle, Xe, metric, qs, k, out[WARM_ALWAYS][0][qs], out[WARM_ALWAYS][1][qs])


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("metric", [capi.METRIC_COSINE, capi.METRIC_NEG_DOT, capi.METRIC_EUCLIDEAN])
def test_admissibility_mask(oracle, metric, path):
    """gorse_topk_set_mask: rows with mask 0 take no part in a search -- the answer is ann.Bruteforce over the admissible rows
    alone (ids unchanged), for a mild and for a 1 %-selective mask (what a category filter looks like), duplicates (ties) among
    the admissible rows included; clearing the mask restores the unfiltered answer."""
    capi.lib().gorse_hip_test_set_topk_path(path)
    rng = np.random.default_rng(31 + metric)
    N, d, k = 6000, 48, 10
    Xf = rng.standard_normal((N, d)).astype(np.float32) * rng.uniform(0.5, 2.0, (N, 1)).astype(np.float32)
    Xf[200:210] = Xf[7]
    t = capi.TopK(Xf, metric)
    plain = t.all_pairs(k)
    qv = rng.standard_normal((80, d)).astype(np.float32)
    for keep in (0.7, 0.01):
        mask = (rng.random(N) < keep).astype(np.uint8)
        mask[200:206] = 1
        mask[7] = 1
        rows = np.nonzero(mask)[0]
        sub = np.ascontiguousarray(Xf[rows])
        t.set_mask(mask)
        idx, dist, cnt = t.search_vector(qv, k)
        for r in range(0, 80, 7):
            ei, ed = oracle.search_vector(sub, metric, qv[r], k)
            assert cnt[r] == ei.size and np.array_equal(idx[r, :cnt[r]], rows[ei]), (keep, r)
            assert np.array_equal(bits(dist[r, :cnt[r]]), bits(ed)), (keep, r)
        qs = rows[::max(1, rows.size // 25)]
        i2, d2, c2 = t.search_index(qs, k)
        for r, q in enumerate(qs):  # exclude_self on a masked index: the query row is one of the admissible rows
            ei, ed = oracle.search_index(sub, metric, int(np.searchsorted(rows, q)), k)
            assert c2[r] == ei.size and np.array_equal(i2[r, :c2[r]], rows[ei]), (keep, q)
            assert np.array_equal(bits(d2[r, :c2[r]]), bits(ed)), (keep, q)
    t.set_mask(None)
    again = t.all_pairs(k)
    assert np.array_equal(again[0], plain[0]) and np.array_equal(bits(again[1]), bits(plain[1]))


SYM_OFF = 1 << 23


@pytest.mark.parametrize("metric", [capi.METRIC_COSINE, capi.METRIC_NEG_DOT, capi.METRIC_EUCLIDEAN])
@pytest.mark.parametrize("dtype,d,norms", [(capi.DTYPE_BF16, 128, "unit"), (capi.DTYPE_BF16, 128, "skewed"), (capi.DTYPE_BF16, 64, "skewed"),
                                           (capi.DTYPE_BF16, 32, "unit"), (capi.DTYPE_F32, 40, "skewed")])
def test_symmetric_sweep_equals_the_square_sweep(oracle, metric, dtype, d, norms):
    """An all-pairs search whose queries are a contiguous range of the stored rows takes the symmetric form of the sweep
    (topk_sweep_kernel<..., SYM>: a workgroup skips the row tiles of later query blocks and reads the tiles of earlier ones along
    their rows, for those rows' foreign lists).  Rows, distance bits and counts must equal the square sweep's (variant bit 23) for
    every query range -- all rows, a range that starts and ends inside the index, one that ends inside a tile -- with every
    epilogue (cosine with equal norms: raw-score thresholds; skewed norms: scaled scores; Euclidean: biased scores), duplicates
    (ties: the history sweep answers them) and a mask; and equal the oracle's on a sample."""
    rng = np.random.default_rng(9000 + 10 * d + metric)
    N, k = 20000 + 77, 30
    Xf = rng.standard_normal((N, d)).astype(np.float32)
    if norms == "unit":
        Xf /= np.linalg.norm(Xf, axis=1, keepdims=True)
    else:
        Xf *= rng.uniform(0.4, 2.5, (N, 1)).astype(np.float32)
    Xf[700:730] = Xf[3]      # duplicates across query blocks
    Xf[15000] = Xf[3]
    X = to_bf16(Xf) if dtype == capi.DTYPE_BF16 else Xf
    Xe = from_bf16(X) if dtype == capi.DTYPE_BF16 else Xf
    t = capi.TopK(X, metric, dtype=dtype)
    for (q0, q1) in ((0, N), (1024, N), (0, 12000), (2560, 17011)):
        capi.lib().gorse_hip_test_set_topk_variant(WARM_ALWAYS | SYM_OFF)
        i_sq, d_sq = t.all_pairs(k, q0, q1)
        assert not t.last_symmetric()
        capi.lib().gorse_hip_test_set_topk_variant(WARM_ALWAYS)
        i_sy, d_sy = t.all_pairs(k, q0, q1)
        assert t.last_symmetric(), (q0, q1)
        n_fb, n_tie = t.last_stats()
        assert n_fb <= 64, "the symmetric sweep handed %d queries to the scan" % n_fb
        assert np.array_equal(i_sy, i_sq), (q0, q1, np.argwhere((i_sy != i_sq).any(1))[:5])
        assert np.array_equal(bits(d_sy), bits(d_sq)), (q0, q1)
        qs = np.concatenate([np.arange(q0, q1, 1777), [q0, q1 - 1]])
        qs = np.unique(np.concatenate([qs, [q for q in (3, 700, 729, 15000) if q0 <= q < q1]])).astype(np.int64)
        check_rows(oracle, Xe, metric, qs, k, i_sy[qs - q0], d_sy[qs - q0])
    # a query range that does not start on a tile boundary takes the square sweep
    t.all_pairs(k, 100, 9000)
    assert not t.last_symmetric()
    # with a mask: masked rows are nobody's candidates, but they are still queries
    mask = (rng.random(N) < 0.6).astype(np.uint8)
    t.set_mask(mask)
    capi.lib().gorse_hip_test_set_topk_variant(WARM_ALWAYS | SYM_OFF)
    i_sq, d_sq = t.all_pairs(k)
    capi.lib().gorse_hip_test_set_topk_variant(WARM_ALWAYS)
    i_sy, d_sy = t.all_pairs(k)
    assert t.last_symmetric()
    assert np.array_equal(i_sy, i_sq) and np.array_equal(bits(d_sy), bits(d_sq))
    assert mask[i_sy[i_sy >= 0]].all()


def test_symmetric_sweep_with_overflowing_foreign_lists(oracle):
    """900 copies of one vector: every copy is a candidate of every other copy, so the foreign lists of the copies overflow and the
    staging areas of the workgroups fill inside one block -- those queries are flagged and answered by the tie path; everybody
    else's rows are untouched."""
    rng = np.random.default_rng(12)
    N, d, k = 24000, 64, 20
    Xf = rng.standard_normal((N, d)).astype(np.float32)
    Xf[5000:5900] = Xf[17]
    Xb = to_bf16(Xf)
    Xe = from_bf16(Xb)
    t = capi.TopK(Xb, capi.METRIC_NEG_DOT, dtype=capi.DTYPE_BF16)
    capi.lib().gorse_hip_test_set_topk_variant(WARM_ALWAYS)
    idx, dist = t.all_pairs(k)
    assert t.last_symmetric()
    qs = np.array([0, 17, 18, 4999, 5000, 5450, 5899, 5900, 12345, N - 1])
    check_rows(oracle, Xe, capi.METRIC_NEG_DOT, qs, k, idx[qs], dist[qs])
