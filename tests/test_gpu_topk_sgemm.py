"""GPU parity: exact top-k (ann.Bruteforce semantics) and floats.MM, through the C ABI."""
import json
import os

import numpy as np
import pytest

from gorse_amd import capi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
KATS = json.load(open(os.path.join(GOLD, "reference_kats.json")))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(autouse=True)
def _reset(oracle):
    oracle.set_isa(orc.ISA_AVX512)
    yield


def test_mf_items_search_golden():
    # logics/cf_test.go:26-58: distance = -Dot, top-3 of 5 vectors for query (1,1,1)
    k = KATS["mf_items_search"]
    X = np.array(k["vectors"], np.float32)
    t = capi.TopK(X, capi.METRIC_NEG_DOT)
    idx, dist, cnt = t.search_vector(np.array(k["query"], np.float32), k["k"])
    assert cnt[0] == 3
    got = [[k["ids"][i], float(-s)] for i, s in zip(idx[0], dist[0])]
    assert got == k["out"]


@pytest.mark.parametrize("metric", [capi.METRIC_NEG_DOT, capi.METRIC_EUCLIDEAN, capi.METRIC_COSINE])
@pytest.mark.parametrize("d", [16, 128, 20, 3])
def test_search_index_exact(oracle, metric, d):
    # ann.Bruteforce.SearchIndex (bruteforce.go:39-63): indices AND distances bit-exact
    rng = np.random.default_rng(d * 10 + metric)
    N, k = 700, 20
    X = rng.standard_normal((N, d)).astype(np.float32)
    t = capi.TopK(X, metric)
    qs = rng.integers(0, N, 25)
    idx, dist, cnt = t.search_index(qs, k)
    for r, q in enumerate(qs):
        ei, ed = oracle.search_index(X, metric, int(q), k)
        assert cnt[r] == ei.size
        assert np.array_equal(idx[r, :cnt[r]], ei)
        assert np.array_equal(bits(dist[r, :cnt[r]]), bits(ed))


def test_search_with_ties_and_prune0(oracle):
    # small-integer vectors => many equal distances: tie order must be Go's container/heap order
    rng = np.random.default_rng(4)
    N, d, k = 300, 8, 15
    X = rng.integers(-2, 3, (N, d)).astype(np.float32)
    for metric in (capi.METRIC_NEG_DOT, capi.METRIC_EUCLIDEAN):
        t = capi.TopK(X, metric)
        qs = np.arange(0, 40)
        for prune0 in (False, True):
            idx, dist, cnt = t.search_index(qs, k, prune0)
            for r, q in enumerate(qs):
                ei, ed = oracle.search_index(X, metric, int(q), k, prune0)
                assert cnt[r] == ei.size, (metric, prune0, q)
                assert np.array_equal(idx[r, :cnt[r]], ei), (metric, prune0, q)
                assert np.array_equal(bits(dist[r, :cnt[r]]), bits(ed))
            assert (idx[np.arange(40)[:, None].repeat(k, 1) >= 0] != -2).all()


def test_search_vector_and_all_pairs(oracle):
    rng = np.random.default_rng(11)
    N, d, k = 500, 32, 10
    X = rng.standard_normal((N, d)).astype(np.float32)
    t = capi.TopK(X, capi.METRIC_NEG_DOT)
    qv = rng.standard_normal((7, d)).astype(np.float32)
    idx, dist, cnt = t.search_vector(qv, k)
    for r in range(7):
        ei, ed = oracle.search_vector(X, orc.METRIC_NEG_DOT, qv[r], k)
        assert np.array_equal(idx[r, :cnt[r]], ei) and np.array_equal(bits(dist[r, :cnt[r]]), bits(ed))
    ai, ad = t.all_pairs(k)
    for q in range(0, N, 37):
        ei, ed = oracle.search_index(X, orc.METRIC_NEG_DOT, q, k)
        assert np.array_equal(ai[q], ei) and np.array_equal(bits(ad[q]), bits(ed))
    assert (ai != np.arange(N)[:, None]).all()  # i != q


def test_bf16_index(oracle):
    # dtype bf16 = truncated fp32 (bfloats.go:24-38), arithmetic in fp32 after <<16
    rng = np.random.default_rng(12)
    N, d, k = 400, 64, 12
    Xf = rng.standard_normal((N, d)).astype(np.float32)
    Xb = (Xf.view(np.uint32) >> 16).astype(np.uint16)
    Xe = (Xb.astype(np.uint32) << 16).view(np.float32)
    for metric in (capi.METRIC_COSINE, capi.METRIC_EUCLIDEAN):
        t = capi.TopK(Xb, metric, dtype=capi.DTYPE_BF16)
        idx, dist, cnt = t.search_index(np.arange(30), k)
        for q in range(30):
            ei, ed = oracle.search_index(Xe, metric, q, k)
            assert np.array_equal(idx[q, :cnt[q]], ei) and np.array_equal(bits(dist[q, :cnt[q]]), bits(ed))


def test_fast_selection_equals_the_literal_heaps():
    """the scan's two selection kernels: select_fast_kernel (queries whose k + 1 smallest distances are pairwise distinct) +
    the literal heap kernel for the rest, against the literal kernel alone -- indices, distance bits, counts, padding; with
    ties (small integers), a mask, prune0, NaN / inf vectors, k + 1 > N, and sizes around the 1024-thread stride"""
    rng = np.random.default_rng(21)
    L = capi.lib()
    cases = []
    for N, d, k in ((5000, 24, 100), (1023, 8, 7), (1025, 8, 300), (40, 5, 64), (3000, 16, 1000)):
        cases.append((rng.standard_normal((N, d)).astype(np.float32), k, None))
    Xi = rng.integers(-2, 3, (2500, 6)).astype(np.float32)  # many equal distances
    cases.append((Xi, 20, None))
    cases.append((Xi, 20, (rng.random(2500) < 0.3).astype(np.uint8)))
    Xn = rng.standard_normal((800, 12)).astype(np.float32)
    Xn[5, 3], Xn[77, 0], Xn[300, 1] = np.nan, np.inf, -np.inf
    cases.append((Xn, 15, None))
    try:
        for X, k, mask in cases:
            for metric in (capi.METRIC_NEG_DOT, capi.METRIC_EUCLIDEAN, capi.METRIC_COSINE):
                t = capi.TopK(X, metric)
                if mask is not None:
                    t.set_mask(mask)
                qs = rng.integers(0, X.shape[0], 9)
                qv = rng.standard_normal((5, X.shape[1])).astype(np.float32)
                for prune0 in (False, True):
                    L.gorse_hip_test_set_scan_literal(0)
                    a = t.search_index(qs, k, prune0) + t.search_vector(qv, k, prune0)
                    L.gorse_hip_test_set_scan_literal(1)
                    b = t.search_index(qs, k, prune0) + t.search_vector(qv, k, prune0)
                    for x, y in zip(a, b):
                        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x,
                                              y.view(np.uint32) if y.dtype == np.float32 else y), (X.shape, k, metric, prune0)
                t.close()
    finally:
        L.gorse_hip_test_set_scan_literal(0)


def test_topk_edge_cases(oracle):
    X = np.arange(12, dtype=np.float32).reshape(4, 3)
    t = capi.TopK(X, capi.METRIC_EUCLIDEAN)
    idx, dist, cnt = t.search_index([0], 10)  # k > N-1
    assert cnt[0] == 3 and (idx[0, 3:] == -1).all() and np.isinf(dist[0, 3:]).all()
    with pytest.raises(capi.GorseHipError) as e:
        t.search_index([4], 2)  # "index out of range" (bruteforce.go:41-43)
    assert e.value.code == capi.ERR_RANGE
    with pytest.raises(capi.GorseHipError):
        t.search_index([-1], 2)
    one = capi.TopK(np.ones((1, 3), np.float32), capi.METRIC_NEG_DOT)
    idx, dist, cnt = one.search_index([0], 2)
    assert cnt[0] == 0


def test_sgemm_goldens():
    # common/floats/floats_test.go:279-301, 425-447
    mm = KATS["floats"]["mm"]
    for c in mm["cases"]:
        out = capi.sgemm(c["transA"], c["transB"], c["m"], c["n"], c["k"], mm["a"], c["lda"], mm["b"], c["ldb"],
                         np.zeros(8, np.float32), c["ldc"])
        assert out.tolist() == c["out"]


def test_sgemm_bit_equal_reference_fixture():
    # tests/golden/ref_simd_vectors.npz: outputs of the reference's own _mm512_mm
    z = np.load(os.path.join(GOLD, "ref_simd_vectors.npz"))
    oa = ob = oc = 0
    for (m, n, k, tA, tB) in z["mm_shapes"]:
        m, n, k = int(m), int(n), int(k)
        a = z["mm_a"][oa:oa + m * k]
        b = z["mm_b"][ob:ob + n * k]
        c0 = z["mm_c0"][oc:oc + m * n]
        lda = m if tA else k
        ldb = k if tB else n
        got = capi.sgemm(tA, tB, m, n, k, a, lda, b, ldb, c0, n)
        assert np.array_equal(bits(got), bits(z["mm_c512"][oc:oc + m * n])), (m, n, k, tA, tB)
        oa += m * k
        ob += n * k
        oc += m * n


def test_sgemm_larger_vs_oracle(oracle):
    rng = np.random.default_rng(2)
    for (m, n, k) in [(65, 130, 77), (128, 128, 128)]:
        for tA in (0, 1):
            for tB in (0, 1):
                a = rng.standard_normal((k, m) if tA else (m, k)).astype(np.float32)
                b = rng.standard_normal((n, k) if tB else (k, n)).astype(np.float32)
                c0 = rng.standard_normal((m, n)).astype(np.float32)
                got = capi.sgemm(tA, tB, m, n, k, a.ravel(), a.shape[1], b.ravel(), b.shape[1], c0.ravel(), n)
                exp = oracle.mm(tA, tB, m, n, k, a.ravel(), a.shape[1], b.ravel(), b.shape[1], c0.ravel(), n)
                assert np.array_equal(bits(got), bits(exp)), (m, n, k, tA, tB)


def test_sgemm_on_the_matrix_cores_is_the_same_chain(oracle):
    """floats.MM's NN / TN / TT cases run on v_mfma_f32_32x32x2_f32 from 64 x 64 results on (csrc/sgemm.hip sgemm_mfma_kernel): every
    element must still be the l-ascending fmaf chain of _mm512_mm onto C's previous value -- the oracle's, bit for bit, and the
    vector-ALU kernel's (test hook).  Shapes off the 128-tile and off the 16-step block, k = 1 / 2 / 3 / odd, padded leading
    dimensions, values over 24 binary orders of magnitude, and a C of signed zeros against all-zero operands (a padded zero step
    would turn -0 into +0: the kernel never takes one)."""
    rng = np.random.default_rng(9)
    L = capi.lib()
    for (m, n, k) in [(64, 64, 1), (64, 65, 2), (70, 64, 3), (129, 257, 33), (300, 131, 130), (257, 300, 67)]:
        for tA, tB in ((0, 0), (1, 0), (1, 1)):
            pad = int(rng.integers(0, 3))
            ar, ac = ((k, m) if tA else (m, k))
            br, bc = ((n, k) if tB else (k, n))
            a = np.ldexp(rng.uniform(-1, 1, (ar, ac + pad)), rng.integers(-12, 12, (ar, ac + pad))).astype(np.float32)
            b = np.ldexp(rng.uniform(-1, 1, (br, bc + pad)), rng.integers(-12, 12, (br, bc + pad))).astype(np.float32)
            c0 = rng.standard_normal((m, n + pad)).astype(np.float32)
            c0[::3, ::5] = -0.0
            args = (tA, tB, m, n, k, a.ravel(), ac + pad, b.ravel(), bc + pad, c0.ravel(), n + pad)
            got = capi.sgemm(*args)
            assert L.gorse_hip_test_sgemm_last_ms() > 0.0
            exp = oracle.mm(*args)
            assert np.array_equal(bits(got), bits(exp)), (m, n, k, tA, tB)
            L.gorse_hip_test_set_sgemm_valu(1)
            try:
                valu = capi.sgemm(*args)
            finally:
                L.gorse_hip_test_set_sgemm_valu(0)
            assert np.array_equal(bits(got), bits(valu)), (m, n, k, tA, tB)
    # signed zeros: C = -0 stays -0 under products that are all +0 or -0 only if no extra step is taken
    m = n = 64
    for k in (4, 5):
        a = np.zeros((m, k), np.float32)
        b = np.zeros((k, n), np.float32)
        b[1::2] = -0.0
        c0 = np.full((m, n), -0.0, np.float32)
        got = capi.sgemm(0, 0, m, n, k, a.ravel(), k, b.ravel(), n, c0.ravel(), n)
        exp = oracle.mm(0, 0, m, n, k, a.ravel(), k, b.ravel(), n, c0.ravel(), n)
        assert np.array_equal(bits(got), bits(exp)), k


def test_sgemm_denormals_on_the_matrix_cores(oracle):
    """Subnormal operands, products that underflow into the subnormal range and chains that climb out of it again: the fp32 MFMA must
    treat them as the fmaf chain of the reference does (no flush to zero anywhere), bit for bit against the oracle and the vector-ALU
    kernel; signed zeros among them."""
    rng = np.random.default_rng(19)
    L = capi.lib()
    m, n, k = 96, 64, 37
    for case in range(3):
        a = np.ldexp(rng.uniform(-1, 1, (m, k)), rng.integers(-80, -60, (m, k))).astype(np.float32)  # ~1e-21: products ~1e-42 (subnormal)
        b = np.ldexp(rng.uniform(-1, 1, (k, n)), rng.integers(-80, -60, (k, n))).astype(np.float32)
        if case == 1:  # subnormal operands times large ones
            a = (a * np.float32(1e-20)).astype(np.float32)
            b = np.ldexp(rng.uniform(-1, 1, (k, n)), rng.integers(0, 30, (k, n))).astype(np.float32)
        if case == 2:  # a chain that starts subnormal and leaves the range
            b[k // 2:] = np.ldexp(rng.uniform(-1, 1, (k - k // 2, n)), 40).astype(np.float32)
        a[::7, ::3] = -0.0
        c0 = np.zeros((m, n), np.float32)
        c0[::2] = np.float32(1e-44)  # a subnormal C
        c0[1::4] = -0.0
        args = (0, 0, m, n, k, a.ravel(), k, b.ravel(), n, c0.ravel(), n)
        got = capi.sgemm(*args)
        exp = oracle.mm(*args)
        if case == 0:
            assert (np.abs(exp[exp != 0]) < 1.2e-38).any()  # subnormal results
        if case == 1:
            assert (np.abs(a[a != 0]) < 1.2e-38).all()  # subnormal operands
        assert np.array_equal(bits(got), bits(exp)), case
        L.gorse_hip_test_set_sgemm_valu(1)
        try:
            valu = capi.sgemm(*args)
        finally:
            L.gorse_hip_test_set_sgemm_valu(0)
        assert np.array_equal(bits(got), bits(valu)), case


def test_sgemm_device_entry_point_equals_the_host_one():
    """gorse_hip_sgemm_device: operands resident in device memory (here torch tensors), C updated in place -- the same bits as the
    host-buffer entry point, on a shape off the tiles and on whole tiles."""
    import torch
    rng = np.random.default_rng(5)
    for (m, n, k), (tA, tB) in (((300, 131, 130), (0, 0)), ((256, 256, 64), (1, 0)), ((64, 65, 3), (1, 1)), ((33, 20, 17), (0, 1))):
        ar, ac = ((k, m) if tA else (m, k))
        br, bc = ((n, k) if tB else (k, n))
        a = rng.standard_normal((ar, ac)).astype(np.float32)
        b = rng.standard_normal((br, bc)).astype(np.float32)
        c0 = rng.standard_normal((m, n)).astype(np.float32)
        want = capi.sgemm(tA, tB, m, n, k, a.ravel(), ac, b.ravel(), bc, c0.ravel(), n)
        ta, tb, tc = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(c0.copy()).cuda()
        torch.cuda.synchronize()
        capi.sgemm_device(tA, tB, m, n, k, ta.data_ptr(), ac, tb.data_ptr(), bc, tc.data_ptr(), n)
        assert np.array_equal(bits(tc.cpu().numpy().ravel()), bits(want)), (m, n, k, tA, tB)


def test_sgemm_errors():
    with pytest.raises(capi.GorseHipError):
        capi.sgemm(0, 0, 2, 2, 2, np.zeros(4), 1, np.zeros(4), 2, np.zeros(4), 2)  # lda too small


@pytest.mark.parametrize("path", [1, 2])
def test_bfloats_euclidean_order(oracle, path):
    """GORSE_METRIC_EUCLIDEAN_BF16: the distance of a bf16 index is bfloats.Euclidean in the summation order of the reference's
    AVX512BW kernel (common/bfloats/src/bfloats_avx512.c:26-59: 16 unfused partials added one after the other, fused scalar
    tail) -- bit for bit against orc_bf16_euclidean (itself pinned to the reference's own kernel, tests/golden/
    ref_simd_vectors.npz), for every vector length class (multiples of 16, tails, shorter than 16), through the literal
    scan (path 1) and the MFMA sweep + exact rescoring (path 2); floats.Euclidean's order gives other bits for the same rows."""
    capi.lib().gorse_hip_test_set_topk_path(path)
    try:
        rng = np.random.default_rng(5)
        differs = 0
        for d in (1, 7, 16, 17, 31, 32, 48, 64, 100, 128, 200):
            N = 700
            Xf = rng.standard_normal((N, d)).astype(np.float32) * rng.uniform(0.2, 3.0, (N, 1)).astype(np.float32)
            Xb = (Xf.view(np.uint32) >> 16).astype(np.uint16)
            Xe = (Xb.astype(np.uint32) << 16).view(np.float32)
            t = capi.TopK(Xb, capi.METRIC_EUCLIDEAN_BF16, dtype=capi.DTYPE_BF16)
            idx, dist = t.all_pairs(20)
            for q in range(0, N, 53):
                ei, ed = oracle.search_index(Xe, orc.METRIC_EUCLIDEAN_BF16, q, 20)
                assert np.array_equal(idx[q], ei), (d, q)
                assert np.array_equal(dist[q].view(np.uint32), ed.view(np.uint32)), (d, q)
                for r in range(0, 20, 7):  # the oracle's distance IS bfloats.Euclidean of the two uint16 rows
                    assert np.float32(oracle.bf16_euclidean(Xb[q], Xb[ei[r]])).view(np.uint32) == ed[r].view(np.uint32)
            plain = capi.TopK(Xb, capi.METRIC_EUCLIDEAN, dtype=capi.DTYPE_BF16).all_pairs(20)[1]
            differs += int((plain.view(np.uint32) != dist.view(np.uint32)).sum())
            qv = (rng.standard_normal((9, d)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
            i2, d2, c2 = t.search_vector(qv, 5)
            for r in range(9):
                ei, ed = oracle.search_vector(Xe, orc.METRIC_EUCLIDEAN_BF16, (qv[r].astype(np.uint32) << 16).view(np.float32), 5)
                assert np.array_equal(i2[r, :c2[r]], ei) and np.array_equal(d2[r, :c2[r]].view(np.uint32), ed.view(np.uint32))
        assert differs > 0  # the two summation orders are not the same function
        with pytest.raises(capi.GorseHipError) as e:
            capi.TopK(Xf, capi.METRIC_EUCLIDEAN_BF16, dtype=capi.DTYPE_F32)
        assert e.value.code == capi.ERR_INVALID
    finally:
        capi.lib().gorse_hip_test_set_topk_path(0)
