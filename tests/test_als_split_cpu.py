"""The three-way split behind the ALS Gram on the bf16 MFMA (gorse_amd/csrc/als.hip gram_accumulate_b3), restated in numpy: a float32
is EXACTLY hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid (round to nearest even, v_cvt_pk_bf16_f32), each of
the three a bf16 value; the six products the kernel forms miss x y by less than 2^-23 |x y| (an fp32 multiply's own rounding: 2^-24)."""
import numpy as np


def split3(x):
    x = np.asarray(x, np.float32)

    def bf16(v):  # round to nearest even (finite inputs)
        u = v.view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)).view(np.float32)

    hi = bf16(x)
    r = x - hi
    mid = bf16(r)
    lo = r - mid
    return hi, mid, lo


def is_bf16(v):
    return bool(((np.asarray(v, np.float32).view(np.uint32) & np.uint32(0xFFFF)) == 0).all())


def test_a_float_is_exactly_three_bf16_values():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(200_000).astype(np.float32) * np.float32(10.0) ** rng.integers(-12, 12, 200_000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 1.0000001, 3.3e38, -3.3e38, 1.2e-30, 2.0 ** -100, 1 + 2.0 ** -23], np.float32)])
    hi, mid, lo = split3(x)
    assert is_bf16(hi) and is_bf16(mid) and is_bf16(lo)
    assert np.array_equal((hi.astype(np.float64) + mid.astype(np.float64)) + lo.astype(np.float64), x.astype(np.float64))
    nz = x != 0
    assert (np.abs(mid[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all() and (np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()


def test_the_six_products_miss_the_product_by_less_than_two_to_the_minus_23():
    rng = np.random.default_rng(4)
    x = rng.standard_normal(300_000).astype(np.float32)
    y = rng.standard_normal(300_000).astype(np.float32)
    xh, xm, xl = (v.astype(np.float64) for v in split3(x))
    yh, ym, yl = (v.astype(np.float64) for v in split3(y))
    six = xm * ym + xh * yl + xl * yh + xh * ym + xm * yh + xh * yh  # every term is exact in the MFMA's fp32 accumulate
    exact = x.astype(np.float64) * y.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -23
