"""The error bounds path B of the exact top-k relies on (gorse_amd/csrc/topk_mfma.hip: topk_mfma_prepare, margin_kernel,
margin_euclid_kernel), re-derived on the CPU: for every (query, candidate) pair the score the MFMA sweep ranks by -- here
the products of its bf16 operands summed without error (float64) and, separately, summed in float32 in two orders, with
the cosine scale / Euclidean bias applied in float32 as the kernel does -- must stay within delta_q of the quantity the
reference's distance defines (oracle = the reference's arithmetic).  The formulas below are the kernels', restated.
Inputs include the nasty regimes: skewed norms, large common offsets (tiny distances between big vectors), d from 3 to 200."""
import numpy as np
import pytest

from oracle import oracle as orc

U = 2.0 ** -24


def bf16_rne(x):
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    b = b + 0x7FFF + ((b >> 16) & 1)
    return ((b >> 16) << 16).astype(np.uint32).view(np.float32)


def operands(X, bf):
    """(candidate operand, query operand) rows: bf16 inputs as they are; fp32 inputs as [hi|lo|hi] / [hi|hi|lo]"""
    if bf:
        return X, X
    hi = bf16_rne(X)
    lo = bf16_rne((X - hi).astype(np.float32))
    return np.concatenate([hi, lo, hi], 1), np.concatenate([hi, hi, lo], 1)


def err_coef(d, bf):
    coef = (3.0 * d + 64.0) * U if bf else (9.0 * d + 64.0) * U + 3.2 * 2.0 ** -16
    return np.float32((coef + 16.0 * U) * 1.01)


def approx_scores(A, Bq):
    """the operand products summed exactly, and in float32 forwards and backwards (any MFMA order lies in this class of
    errors: gamma_K |a|.|b|)"""
    prods = A.astype(np.float64) * Bq.astype(np.float64)[None, :]
    exact = prods.sum(1)
    p32 = prods.astype(np.float32)
    fwd = np.cumsum(p32, axis=1, dtype=np.float32)[:, -1].astype(np.float64)
    bwd = np.cumsum(p32[:, ::-1], axis=1, dtype=np.float32)[:, -1].astype(np.float64)
    return [exact, fwd, bwd]


def make(rng, n, d, kind):
    X = rng.standard_normal((n, d)).astype(np.float32)
    if kind == "skewed":
        X *= rng.lognormal(0, 1.5, (n, 1)).astype(np.float32)
    elif kind == "offset":  # big common component, small differences
        X = (X * 0.01 + 37.0 * rng.standard_normal((1, d))).astype(np.float32)
    elif kind == "tiny":
        X *= np.float32(1e-4)
    return X


@pytest.mark.parametrize("bf", [True, False])
@pytest.mark.parametrize("kind", ["plain", "skewed", "offset", "tiny"])
@pytest.mark.parametrize("d", [3, 64, 128, 200])
def test_sweep_scores_stay_within_their_bounds(oracle, d, kind, bf):
    oracle.set_isa(orc.ISA_AVX512)
    rng = np.random.default_rng(d * 7 + len(kind) + int(bf))
    n = 160
    X = make(rng, n, d, kind)
    if bf:
        X = (X.view(np.uint32) & 0xFFFF0000).view(np.float32)  # a bf16 index: truncated inputs, exact in the operands
    A, B = operands(X, bf)
    norm2 = np.array([-oracle.distance(orc.METRIC_NEG_DOT, x, x) for x in X], np.float32)  # floats.Dot(x, x)
    xmax = np.float32(np.sqrt(norm2.max()) * 1.0001)
    coef = err_coef(d, bf)
    worst = {"dot": 0.0, "cos": 0.0, "l2": 0.0}
    for q in range(0, n, 16):
        qn = np.float32(np.sqrt(norm2[q]))
        sc = approx_scores(A, B[q])
        ref_dot = np.array([-oracle.distance(orc.METRIC_NEG_DOT, X[q], x) for x in X], np.float64)
        ref_cos = np.array([oracle.distance(orc.METRIC_COSINE, X[q], x) for x in X], np.float64)
        ref_l2 = np.array([oracle.distance(orc.METRIC_EUCLIDEAN, X[q], x) for x in X], np.float64)
        # -dot: margin_kernel with other = max norm
        delta = float(coef) * float(qn) * float(xmax) * 1.001 + 1e-30
        for s in sc:
            worst["dot"] = max(worst["dot"], float(np.max(np.abs(s - ref_dot)) / delta))
        # cosine: the sweep multiplies by 1 / |x| in float32 and compares with (1 - distance) * |q|; other = 1
        rs = (np.float32(1.0) / np.sqrt(norm2)).astype(np.float32)
        delta = float(coef) * float(qn) * 1.001 + 1e-30
        target = (1.0 - ref_cos) * float(qn)
        for s in sc:
            got = (s.astype(np.float32) * rs).astype(np.float64)
            worst["cos"] = max(worst["cos"], float(np.max(np.abs(got - target)) / delta))
        # Euclidean: score = q.x - |x|^2 / 2 against (|q|^2 - d_ref^2) / 2 (margin_euclid_kernel)
        du = np.float32(d * 5.9604645e-8)
        u32 = np.float32(5.9604645e-8)
        delta = float(coef * qn * xmax + np.float32(0.5) * (du + 4 * u32) * xmax * xmax
                      + np.float32(0.5) * (du + 8 * u32) * (qn + xmax) * (qn + xmax))
        bias = (np.float32(-0.5) * norm2).astype(np.float32)
        qq_exact = float(np.dot(X[q].astype(np.float64), X[q].astype(np.float64)))
        target = (qq_exact - ref_l2 ** 2) / 2.0
        for s in sc:
            got = (s.astype(np.float32) + bias).astype(np.float64)
            worst["l2"] = max(worst["l2"], float(np.max(np.abs(got - target)) / delta))
    print("worst |score error| / delta: d=%d %s bf=%s %s" % (d, kind, bf, {k: round(v, 4) for k, v in worst.items()}))
    assert worst["dot"] <= 1.0, worst
    assert worst["cos"] <= 1.0, worst
    assert worst["l2"] <= 1.0, worst
