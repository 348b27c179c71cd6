"""The triangle-sharded all-pairs search (include/gorse_hip.h gorse_topk_tri_*; gorse_amd/dist.py refresh_neighbors_triangle*): `world`
ranks EMULATED on the one device -- a handle per rank, rank r's query blocks swept as pass r, thresholds and foreign candidate lists
exchanged through host memory (the stand-in for the all-gather / all-to-all a real node runs) -- must return, for EVERY row, the
indices and distance bits of the single-rank pass (common/ann/bruteforce.go:39-83 through the symmetric sweep), for world 2, 3, 4, 8."""
import time

import numpy as np
import pytest

from gorse_amd import capi, synth
from gorse_amd import dist as gdist
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def run_world(handles, k, world, fetch=True):
    tm = {}
    t0 = time.perf_counter()
    idx, dist = gdist.refresh_neighbors_triangle_local([gdist.HipTriEngine(h, fetch=fetch) for h in handles[:world]], k, timings=tm)
    tm["wall"] = time.perf_counter() - t0
    return idx, dist, tm


@pytest.mark.parametrize("metric,dtype,d", [(capi.METRIC_COSINE, capi.DTYPE_BF16, 32), (capi.METRIC_NEG_DOT, capi.DTYPE_F32, 40),
                                            (capi.METRIC_EUCLIDEAN, capi.DTYPE_BF16, 64)])
def test_triangle_shards_equal_the_single_rank_pass(oracle, metric, dtype, d):
    # (fp32 rows travel as three bf16 parts: d = 40 -> 120 operand columns, the deepest fp32 width with a symmetric sweep)
    N, k = 140_000 + 77, 20  # a partial last block (and tile); >= 2^17 rows: the sweep is warm-started
    Xb, Xe = synth.s_emb(N, d, 91)
    if dtype == capi.DTYPE_F32:
        rng = np.random.default_rng(5)
        X = (Xe * rng.uniform(0.5, 2.0, (N, 1))).astype(np.float32)  # unequal norms
    else:
        X = Xb
        Xb[5000:5200] = Xb[100:300]  # duplicate rows: ties among the best k + 1 -> the tie path
        Xe[5000:5200] = Xe[100:300]
    handles = [capi.TopK(X, metric, dtype=dtype) for _ in range(4)]
    ref_i, ref_d = handles[0].all_pairs(k)
    # (the single-rank pass itself may take the square sweep -- where the pilot leaves more than 1/32 of the queries without a
    # threshold: the tie-heavy bf16 case -- the triangle shards never do: those queries take the tie path, the rows are the same)
    print("single-rank pass symmetric: %s, tie path %d queries" % (handles[0].last_symmetric(), handles[0].last_stats()[1]))
    for world in (1, 2, 3, 4):
        idx, dist, tm = run_world(handles, k, world)
        assert np.array_equal(idx, ref_i), "world %d: indices differ in %d rows" % (world, int((idx != ref_i).any(axis=1).sum()))
        assert np.array_equal(bits(dist), bits(ref_d)), "world %d: distance bits differ" % world
        assert all(h.last_symmetric() for h in handles[:world])
    # the long blocks cut by rows (own lists per slice; chosen by the launch's size -- at this size never): forced to 2 and 4 slices
    L = capi.lib()
    try:
        for forced, world in ((2, 1), (3, 2), (2, 3)):
            L.gorse_hip_test_set_topk_variant(forced << 29)
            idx, dist, _ = run_world(handles, k, world)
            assert np.array_equal(idx, ref_i) and np.array_equal(bits(dist), bits(ref_d)), "slices forced (%d), world %d" % (forced, world)
    finally:
        L.gorse_hip_test_set_topk_variant(0)
    # the one-call form for a process that holds every rank's handle (gorse_topk_tri_all_pairs_local: what the Go master calls)
    for world in (2, 4):
        li, ld = capi.topk_tri_all_pairs_local(handles[:world], k)
        assert np.array_equal(li, ref_i) and np.array_equal(bits(ld), bits(ref_d)), "one-call form, world %d" % world
    # a sample against the oracle itself (the single-rank pass is checked against it elsewhere: tests/test_gpu_topk_mfma.py)
    o_metric = {capi.METRIC_COSINE: orc.METRIC_COSINE, capi.METRIC_NEG_DOT: orc.METRIC_NEG_DOT, capi.METRIC_EUCLIDEAN: orc.METRIC_EUCLIDEAN}[metric]
    Xo = X if dtype == capi.DTYPE_F32 else Xe
    for q in (0, 511, 512, 5100, N - 1):
        ei, ed = oracle.search_index(Xo, o_metric, q, k)
        assert np.array_equal(idx[q], ei) and np.array_equal(bits(dist[q]), bits(ed)), q


def test_triangle_calls_out_of_order_and_ineligible_searches_fail_loudly():
    N, d = 140_000, 32
    Xb, _ = synth.s_emb(N, d, 92)
    t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
    with pytest.raises(capi.GorseHipError):
        t.tri_sweep()  # nothing begun
    with pytest.raises(capi.GorseHipError):
        t.tri_begin(10, 2, 2)  # rank outside the world
    with pytest.raises(capi.GorseHipError):
        t.tri_begin(10, 0, 2, q_begin=100)  # not on a 128-row boundary: no symmetric form -> shard the query rows
    t.tri_begin(10, 0, 2, q_begin=128)
    with pytest.raises(capi.GorseHipError):
        t.tri_finish()  # not swept yet
    with pytest.raises(capi.GorseHipError):
        t.tri_pack(1)
    small = capi.TopK(Xb[:50_000], capi.METRIC_COSINE, dtype=capi.DTYPE_BF16)
    with pytest.raises(capi.GorseHipError):
        small.tri_begin(10, 0, 2)  # fewer than 2^17 rows: the sweep is not warm-started


def test_c4_triangle_shards_emulated_on_one_device(oracle):
    """BASELINE config C4 (S-emb 1M x 128 bf16, cosine, k = 100) with the triangle sharded over 2, 4 and 8 EMULATED ranks: all
    1,000,000 rows equal the single-rank symmetric pass in indices and distance bits; per rank the seconds of its stages are
    printed (DESIGN.md section 5's "measured on one GPU" column: a rank's pass = pilots + sweep + pack + unpack + finish)."""
    N, d, k = 1_000_000, 128, 100
    Xb, Xe = synth.s_emb(N, d, 44)
    handles = [capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16) for _ in range(8)]
    handles[0].all_pairs(k, fetch=False)
    t0 = time.perf_counter()
    ref_i, ref_d = handles[0].all_pairs(k)
    t_single = time.perf_counter() - t0
    assert handles[0].last_symmetric()
    run_world(handles, k, 8)  # first use of every handle: allocations
    handles[0].all_pairs(k, fetch=False)
    handles[0].synchronize()
    t0 = time.perf_counter()
    handles[0].all_pairs(k, fetch=False)
    handles[0].synchronize()
    t_resident = time.perf_counter() - t0
    for world in (2, 4, 8):
        idx, dist, tm = run_world(handles, k, world)
        bad = int(((idx != ref_i) | (bits(dist) != bits(ref_d))).any(axis=1).sum())
        assert bad == 0, "world %d: %d rows differ from the single-rank pass" % (world, bad)
        # the same once more with the results left on the device (what bench.py times): per rank the device stages alone
        classes = (capi.PROF_TOPK_SWEEP, capi.PROF_TOPK_SELECT, capi.PROF_TOPK_HIST, capi.PROF_TOPK_REPLAY)
        for h in handles[:world]:
            h.set_profiling(True)
        before = [[h.get_profile(c)[1] for c in classes] for h in handles[:world]]
        _, _, tm = run_world(handles, k, world, fetch=False)
        kern = np.array([[h.get_profile(c)[1] for c in classes] for h in handles[:world]]) - np.array(before)
        for h in handles[:world]:
            h.set_profiling(False)
        print("   kernels per rank (hipEvents inside the library), mean ms: pilots + main sweep %.1f, rescoring %.1f, tie path: history sweep %.1f, "
              "sort + replay %.1f; tie-path queries per rank %s" % (*kern.mean(axis=0).tolist(), [h.last_stats()[1] for h in handles[:world]]))
        per = np.array(tm["per_rank_seconds"]) * 1e3
        dev = per[:, [0, 1, 2, 4]].sum(axis=1)  # pilots + sweep + pack kernels + finish; the copies of the messages apart
        mb = sum(tm["message_bytes"].values()) / 1e6
        print("C4 triangle, world %d: all 1,000,000 rows equal the single-rank pass; per rank ms [pilots, sweep, pack, unpack + H2D, finish, "
              "message D2H] mean %s; a rank's device stages (pilots + sweep + pack + finish): max %.1f mean %.1f ms, the single-rank pass "
              "%.1f ms (results resident); foreign lists exchanged %.0f MB in all"
              % (world, np.round(per.mean(axis=0), 1).tolist(), dev.max(), dev.mean(), t_resident * 1e3, mb))
    rows = [0, 511, 999_999, 123_456]
    for q in rows:
        ei, ed = oracle.search_index(Xe, orc.METRIC_COSINE, q, k)
        assert np.array_equal(idx[q], ei) and np.array_equal(bits(dist[q]), bits(ed)), q
