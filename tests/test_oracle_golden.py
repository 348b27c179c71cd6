"""Pin the CPU oracle: (1) the reference's own known-answer tests, (2) bit-equality
with the reference's SIMD C kernels (committed fixture + live oracle/_ref when present),
(3) published Philox4x32-10 known-answer vectors."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(GOLD, "reference_kats.json")))
ALL_ISA = [orc.ISA_GO, orc.ISA_AVX, orc.ISA_AVX512]


@pytest.fixture(autouse=True)
def _reset(oracle):
    oracle.set_isa(orc.ISA_AVX512)
    oracle.set_exp(0)
    yield
    oracle.set_isa(orc.ISA_AVX512)
    oracle.set_exp(0)


@pytest.mark.parametrize("isa", ALL_ISA)
def test_floats_kats(oracle, isa):
    # common/floats/floats_test.go:58-179 -- integers, so every ISA path is exact
    oracle.set_isa(isa)
    f = KATS["floats"]
    assert oracle.mul_const_to(f["mul_const_to"]["a"], f["mul_const_to"]["c"]).tolist() == f["mul_const_to"]["out"]
    k = f["mul_const_add"]
    assert oracle.mul_const_add(k["a"], k["c"], k["dst"]).tolist() == k["out"]
    k = f["mul_const_add_to"]
    assert oracle.mul_const_add_to(k["a"], k["b"], k["c"]).tolist() == k["out"]
    k = f["sub_to"]
    assert oracle.sub_to(k["a"], k["b"]).tolist() == k["out"]
    assert oracle.dot(f["dot"]["a"], f["dot"]["b"]) == f["dot"]["out"]
    assert np.float32(oracle.euclidean(f["euclidean"]["a"], f["euclidean"]["b"])) == np.float32(f["euclidean"]["out"])


@pytest.mark.parametrize("isa", ALL_ISA)
def test_mm_kats(oracle, isa):
    # floats_test.go:279-301, 425-447
    oracle.set_isa(isa)
    mm = KATS["floats"]["mm"]
    for c in mm["cases"]:
        out = oracle.mm(c["transA"], c["transB"], c["m"], c["n"], c["k"], mm["a"], c["lda"], mm["b"], c["ldb"],
                        np.zeros(8, np.float32), c["ldc"])
        assert out.tolist() == c["out"]


@pytest.mark.parametrize("isa", ALL_ISA)
def test_bfloats_kats(oracle, isa):
    # common/bfloats/bfloats_test.go:23-28 ; truncation bfloats.go:24-30
    oracle.set_isa(isa)
    k = KATS["bfloats"]["euclidean"]
    a = oracle.bf16_from_f32(k["a"])
    b = oracle.bf16_from_f32(k["b"])
    assert np.float32(oracle.bf16_euclidean(a, b)) == np.float32(k["out"])
    x = np.array([0.1, 0.2, 0.3, -1.7, 3.1415927], np.float32)
    assert oracle.bf16_from_f32(x).tolist() == (x.view(np.uint32) >> 16).tolist()
    assert oracle.bf16_to_f32(oracle.bf16_from_f32(x)).view(np.uint32).tolist() == \
        ((x.view(np.uint32) >> 16) << 16).tolist()


def test_heap_kats(oracle):
    # common/heap/filter_test.go:22-45, pq_test.go:28-61
    for k in KATS["heap"]["topk_filter"]:
        v, w = oracle.topk_filter(k["k"], k["items"], k["weights"])
        assert v.tolist() == k["values"] and w.tolist() == k["out_weights"]
    pq = KATS["heap"]["priority_queue"]
    e = pq["elements"]
    v, w = oracle.pq_sort(False, e, e)
    assert v.tolist() == pq["asc"] and w.tolist() == pq["asc"]
    v, w = oracle.pq_sort(True, e, e)
    assert v.tolist() == pq["desc"]


def test_metric_kats(oracle):
    # model/cf/evaluator_test.go:33-74
    m = KATS["metrics"]
    ids = {"ndcg": orc.M_NDCG, "precision": orc.M_PRECISION, "recall": orc.M_RECALL, "hr": orc.M_HR,
           "map": orc.M_MAP, "mrr": orc.M_MRR}
    for c in m["cases"]:
        assert abs(oracle.metric(ids[c["metric"]], c["target"], m["rank"]) - c["out"]) < m["epsilon"]


def test_evaluate_kat(oracle):
    # model/cf/evaluator_test.go:137-171: mock scores expressed as rank-1 factors is impossible,
    # so run Rank/Precision through the same TopKFilter path on the mock's score table.
    e = KATS["evaluate"]
    total = np.float32(0)
    for u in range(4):
        test_items = [i for i in range(16) if i // 4 == u]
        negs = [i for i in range(16) if i // 4 != u]       # SampleUserNegatives with n >= range
        cand = test_items + negs
        score = [1.0 if i in e["positive"][u] else (-1.0 if i in e["negative"][u] else 0.0) for i in cand]
        rank, _ = oracle.topk_filter(e["topk"], cand, score)
        total += np.float32(oracle.metric(orc.M_PRECISION, test_items, rank))
    assert np.float32(total * np.float32(1 / np.float32(4))) == np.float32(e["precision"])


def test_mf_items_search_kat(oracle):
    # logics/cf_test.go:26-58 with distance = -floats.Dot (logics/cf.go:32-34)
    k = KATS["mf_items_search"]
    X = np.array(k["vectors"], np.float32)
    idx, dist = oracle.search_vector(X, orc.METRIC_NEG_DOT, k["query"], k["k"])
    got = [[k["ids"][i], float(-d)] for i, d in zip(idx, dist)]
    assert got == k["out"]


def _check_fixture(oracle, getter):
    z = np.load(os.path.join(GOLD, "ref_simd_vectors.npz"))
    off = 0
    for t, n in enumerate(z["lengths"]):
        n = int(n)
        a, b, c = z["a"][off:off + n], z["b"][off:off + n], z["c"][off:off + n]
        s = float(z["s"][t])
        exp = getter(z, t, off, n)
        for isa, keys in ((orc.ISA_AVX512, ("dot512", "euc512", "bfeuc512", "mca512")),
                          (orc.ISA_AVX, ("dot256", "euc256", "bfeuc256", "mca256"))):
            oracle.set_isa(isa)
            if n > 0:
                assert np.float32(oracle.dot(a, b)).view(np.uint32) == exp[keys[0]].view(np.uint32), (n, isa)
                assert np.float32(oracle.euclidean(a, b)).view(np.uint32) == exp[keys[1]].view(np.uint32), (n, isa)
                ab = (a.view(np.uint32) >> 16).astype(np.uint16)
                bb = (b.view(np.uint32) >> 16).astype(np.uint16)
                assert np.float32(oracle.bf16_euclidean(ab, bb)).view(np.uint32) == exp[keys[2]].view(np.uint32), (n, isa)
            assert np.array_equal(oracle.mul_const_add(a, s, c).view(np.uint32), exp[keys[3]].view(np.uint32)), (n, isa)
        oracle.set_isa(orc.ISA_AVX512)
        assert np.array_equal(oracle.mul_const_add_to(a, s, c).view(np.uint32), exp["mcat512"].view(np.uint32))
        off += n


def test_bit_equal_to_reference_kernels_fixture(oracle):
    """Oracle == the reference's compiled C kernels, bit for bit (fixture made by
    scripts/gen_golden_ref_vectors.py from /root/reference)."""
    def getter(z, t, off, n):
        d = {k: z[k][t] for k in ("dot512", "dot256", "euc512", "euc256", "bfeuc512", "bfeuc256")}
        d["mca512"] = z["mca512"][off:off + n]
        d["mca256"] = z["mca256"][off:off + n]
        d["mcat512"] = z["mcat512"][off:off + n]
        return d
    _check_fixture(oracle, getter)


def test_mm_bit_equal_fixture(oracle):
    z = np.load(os.path.join(GOLD, "ref_simd_vectors.npz"))
    oa = ob = oc = 0
    for (m, n, k, tA, tB) in z["mm_shapes"]:
        m, n, k = int(m), int(n), int(k)
        a = z["mm_a"][oa:oa + m * k]
        b = z["mm_b"][ob:ob + n * k]
        c0 = z["mm_c0"][oc:oc + m * n]
        lda = m if tA else k
        ldb = k if tB else n
        for isa, key in ((orc.ISA_AVX512, "mm_c512"), (orc.ISA_AVX, "mm_c256")):
            oracle.set_isa(isa)
            got = oracle.mm(tA, tB, m, n, k, a, lda, b, ldb, c0, n)
            assert np.array_equal(got.view(np.uint32), z[key][oc:oc + m * n].view(np.uint32)), (m, n, k, tA, tB, isa)
        oa += m * k
        ob += n * k
        oc += m * n


def test_bit_equal_to_live_reference_kernels(oracle, ref):
    """Same check against oracle/_ref run live, on fresh random data."""
    if ref is None:
        pytest.skip("oracle/_ref/libgorse_ref.so absent or host lacks AVX512")
    rng = np.random.default_rng(7)
    for n in list(range(1, 140)) + [255, 256, 257]:
        a = rng.standard_normal(n).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        for isa in (orc.ISA_AVX, orc.ISA_AVX512):
            oracle.set_isa(isa)
            assert np.float32(oracle.dot(a, b)).view(np.uint32) == np.float32(ref.dot(isa, a, b)).view(np.uint32)
            assert np.float32(oracle.euclidean(a, b)).view(np.uint32) == \
                np.float32(ref.euclidean(isa, a, b)).view(np.uint32)


def test_simd_equals_scalar_on_integers(oracle):
    # the reference's SIMDTestSuite (floats_test.go:307-423): every ISA == pure Go on small integers
    a = np.arange(20, dtype=np.float32)
    b = np.arange(20, dtype=np.float32) * 2
    outs = []
    for isa in ALL_ISA:
        oracle.set_isa(isa)
        outs.append((oracle.dot(a, b), oracle.euclidean(a, b), oracle.mul_const_add(a, 2.0, b).tolist()))
    assert outs[0] == outs[1] == outs[2]


def test_philox_kat(oracle):
    # Random123 kat_vectors, philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11)
    assert oracle.philox([0, 0, 0, 0], [0, 0]).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2).tolist() == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]).tolist() == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_exp_restated_close_to_libm(oracle):
    xs = np.linspace(-20, 20, 4001).astype(np.float32)
    for x in xs:
        r = oracle.L.orc_exp_restated(float(x))
        e = np.exp(np.float64(x))
        assert abs(r - e) <= 2e-7 * e


# ---- sparse collections: what the reference's tests pin (rank order on distinct scores, disjoint vectors left out) ----
def test_sparse_search_reference_test_sparse(oracle):
    # storage/vectors/database_test.go:198-224: old {1:1,100:1}, match {1:1,100:2}, other {2:1,200:2}; query {1:1,100:2}
    ptr = np.array([0, 2, 4, 6], np.int64)
    idx = np.array([1, 100, 1, 100, 2, 200], np.uint32)
    val = np.array([1, 1, 1, 2, 1, 2], np.float32)
    i, s = oracle.sparse_search(ptr, idx, val, [1, 100], [1, 2], 10)
    assert i.tolist() == [1, 0] and s.tolist() == [5.0, 3.0]  # "match" first, "other" (no common index) not returned


def _nested(n_items, idf):
    """item_to_item_test.go:244-270 (TestUsers): item i has feedback from users 1 .. 100-i; every idf = 1"""
    rows = [np.arange(1, n_items - i + 1) for i in range(n_items)]
    ptr = np.array([0] + list(np.cumsum([r.size for r in rows])), np.int64)
    idx = np.concatenate(rows).astype(np.uint32)
    val = np.sqrt(np.asarray(idf, np.float64)[idx]).astype(np.float32)  # vector_writer.go:206
    return ptr, idx, val


def test_sparse_search_reference_test_users_rank_order(oracle):
    ptr, idx, val = _nested(100, np.ones(101))
    i, s = oracle.sparse_search(ptr, idx, val, idx[ptr[0]:ptr[1]], val[ptr[0]:ptr[1]], 11)
    # QueryItemToItem asks for n + 1 neighbours and drops the item itself: "1" .. "10" (item_to_item_test.go:266-270)
    assert i.tolist() == list(range(0, 11)) and s.tolist() == [float(100 - t) for t in range(11)]


def test_sparse_dot_order_and_ties(oracle):
    common, s = oracle.sparse_dot([1, 3, 9], [1.0, 2.0, 3.0], [3, 9, 11], [0.5, 2.0, 7.0])
    assert common == 2 and s == np.float32(7.0)
    # float32 accumulation in ascending index order: (1e8 + 1) - 1e8 is 0 in that order, not 1
    common, s = oracle.sparse_dot([0, 1, 2], [1e8, 1.0, -1e8], [0, 1, 2], [1.0, 1.0, 1.0])
    assert common == 3 and s == 0.0
    # equal scores come back in ascending row order; a zero score (cancelled products, or no common index) takes a slot
    # of the top k and is then dropped (xvec.go:419-421), so the negative row 3 only shows once rows 0, 1, 2, 4 fit too
    ptr = np.array([0, 1, 2, 4, 5, 6], np.int64)
    idx = np.array([5, 5, 5, 6, 6, 9], np.uint32)
    val = np.array([2.0, 2.0, 1.0, -1.0, -3.0, 1.0], np.float32)
    for k, rows in ((2, [0, 1]), (4, [0, 1]), (5, [0, 1, 3])):
        i, sc = oracle.sparse_search(ptr, idx, val, [5, 6], [1.0, 1.0], k)
        assert i.tolist() == rows and sc.tolist() == [2.0, 2.0, -3.0][:len(rows)]
    i, sc = oracle.sparse_search(ptr, idx, val, [5, 6], [1.0, 1.0], 5, exclude=0, admissible=[1, 0, 1, 1, 1])
    assert i.tolist() == [3]  # rows 2 and 4 score zero, 3 admissible rows < k


def test_idf_formula(oracle):
    # dataset/dataset.go:160-166: math32.Log(1 + float32(n) / float32(freq))
    out = oracle.idf([1, 2, 50, 100], 100)
    exp = [np.float32(np.log(np.float64(np.float32(1) + np.float32(100) / np.float32(f)))) for f in (1, 2, 50, 100)]
    assert out.tolist() == exp


def test_sparse_dot_both_search_strategies_equal_a_plain_sequential_sum(oracle):
    """orc_sparse_dot merges similar-length lists and binary-searches lopsided ones: both must give the products of the
    common indices summed in ascending index order in float32, written here as the plainest possible loop"""
    rng = np.random.default_rng(0)
    for _ in range(400):
        D = int(rng.integers(1, 400))
        na, nb = int(rng.integers(0, min(D, 300) + 1)), int(rng.integers(0, min(D, 30) + 1))
        ia = np.sort(rng.choice(D, na, replace=False)).astype(np.uint32)
        ib = np.sort(rng.choice(D, nb, replace=False)).astype(np.uint32)
        va, vb = rng.standard_normal(na).astype(np.float32), rng.standard_normal(nb).astype(np.float32)
        da, db = dict(zip(ia.tolist(), va)), dict(zip(ib.tolist(), vb))
        s = np.float32(0)
        common = np.intersect1d(ia, ib)
        for t in common:
            s = np.float32(s + np.float32(da[int(t)] * db[int(t)]))
        for x, y in (((ia, va), (ib, vb)), ((ib, vb), (ia, va))):
            c, got = oracle.sparse_dot(x[0], x[1], y[0], y[1])
            assert c == common.size and np.float32(got).view(np.uint32) == s.view(np.uint32)


def test_inverted_index_search_equals_the_brute_force_search(oracle):
    """orc_sparse_search_inverted (the CPU baseline bench.py times: posting lists + accumulators) returns what
    orc_sparse_search (every row merged against the query) returns -- rows, score bits, counts -- with masks, exclusions,
    cancelling and negative scores, k below and above the number of hits"""
    from sparse_cases import random_csr, tie_case, TIE_EXPECT
    rng = np.random.default_rng(9)
    for case in range(12):
        rows, dims = int(rng.integers(1, 500)), int(rng.integers(1, 60))
        ptr, idx, val = random_csr(rng, rows, dims, 0, min(dims, 12), neg=bool(case % 2), zipf=bool(case % 3))
        ix = oracle.sparse_index(ptr, idx, val)
        scratch = ix.scratch()
        mask = (rng.random(rows) < 0.7).astype(np.uint8) if case % 4 == 1 else None
        for q in range(min(rows, 25)):
            qi, qv = idx[ptr[q]:ptr[q + 1]], val[ptr[q]:ptr[q + 1]]
            for k in (1, 5, 40, 600):
                ex = q if k != 5 else -1
                ei, es = oracle.sparse_search(ptr, idx, val, qi, qv, k, exclude=ex, admissible=mask)
                gi, gs, walked = ix.search(qi, qv, k, exclude=ex, admissible=mask, scratch=scratch)
                assert gi.tolist() == ei.tolist() and gs.view(np.uint32).tolist() == es.view(np.uint32).tolist()
        assert not scratch[0].any() and not scratch[1].any()  # the scratch comes back clean
    ptr, idx, val, (qp, qi, qv), mask, excl = tie_case()
    ix = oracle.sparse_index(ptr, idx, val)
    for k, expect in TIE_EXPECT.items():
        gi, gs, walked = ix.search(qi[:2], qv[:2], k, exclude=0, admissible=mask)
        assert gi.tolist() == expect


def test_metric_3_is_bfloats_euclidean(oracle):
    """orc_distance(3, ...) on << 16-expanded rows = orc_bf16_euclidean on the uint16 rows (the kernel that
    tests/golden/ref_simd_vectors.npz pins to the reference's own bfloats_avx512.c)"""
    rng = np.random.default_rng(2)
    for n in (0, 1, 15, 16, 17, 64, 100, 200):
        a = (rng.standard_normal(n).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        b = (rng.standard_normal(n).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        af, bf = (a.astype(np.uint32) << 16).view(np.float32), (b.astype(np.uint32) << 16).view(np.float32)
        assert np.float32(oracle.distance(orc.METRIC_EUCLIDEAN_BF16, af, bf)).view(np.uint32) == \
            np.float32(oracle.bf16_euclidean(a, b)).view(np.uint32)
