"""The integer encodings the ranking kernels sort and compare by (gorse_amd/csrc/rank_keys.hpp -- the header the kernels
include), checked on the CPU through the host library's hook gh_test_rank_key: order preservation, round trips, the -0
rules of each key, and the count of results the reference returns from a sparse search (xvec.go:379-446)."""
import ctypes as C
import itertools

import numpy as np
import pytest

from gorse_amd import cf


@pytest.fixture(scope="module")
def L():
    lib = C.CDLL(cf.HOST_LIB)
    lib.gh_test_rank_key.argtypes = [C.c_int32, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_uint32)]
    lib.gh_test_rank_key.restype = None
    lib.gh_test_sparse_key.argtypes = [C.c_float, C.c_int32]
    lib.gh_test_sparse_key.restype = C.c_uint64
    lib.gh_test_sparse_written.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int32]
    lib.gh_test_sparse_written.restype = C.c_int32
    return lib


def keys(L, what, x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.size, np.uint32)
    L.gh_test_rank_key(what, x.ctypes.data_as(C.POINTER(C.c_float)), x.size, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def samples():
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 1 << 32, 200_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[~np.isnan(x)]
    special = np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 1.17549435e-38, -1.17549435e-38, 3.4028235e38, -3.4028235e38, 1.0, -1.0],
                       np.float32)
    return np.concatenate([x, special, rng.standard_normal(50_000).astype(np.float32)])


def test_fkey_orders_every_bit_pattern_and_round_trips(L):
    x = samples()
    k = keys(L, 0, x)
    order = np.argsort(x, kind="stable")
    xs, ks = x[order], k[order].astype(np.int64)
    assert (np.diff(ks)[np.diff(xs) > 0] > 0).all()  # strictly larger float -> strictly larger key
    assert keys(L, 0, np.float32([-0.0]))[0] < keys(L, 0, np.float32([0.0]))[0]  # distinct patterns stay distinct: -0 below +0
    assert np.array_equal(keys(L, 1, x), x.view(np.uint32))  # fkey_inv(fkey(x)) is x, bit for bit


def test_dist_key_is_ascending_merges_the_zeros_and_gives_the_bits_back(L):
    x = samples()
    k = keys(L, 2, x).astype(np.int64)
    order = np.argsort(x, kind="stable")
    xs, ks = x[order], k[order]
    assert (np.diff(ks)[np.diff(xs) > 0] > 0).all() and (np.diff(ks)[np.diff(xs) == 0] == 0).all()  # equal floats (incl. +-0) = equal keys
    assert (k < 0xFFFFFFFF).all()  # the sentinel of the query itself / the padding is never a distance's key
    assert np.array_equal(keys(L, 3, x), x.view(np.uint32))  # the distance comes back with its own bits, -0 included
    nonpos = ~(x > 0)
    assert np.array_equal(k <= 0x80000000, nonpos)  # prune0's test on the key


def test_sparse_key_ranks_by_score_then_ascending_row(L):
    x = samples()
    o = keys(L, 4, x).astype(np.int64)
    order = np.argsort(x, kind="stable")
    assert (np.diff(o[order])[np.diff(x[order]) > 0] > 0).all()
    assert keys(L, 4, np.float32([-0.0]))[0] == keys(L, 4, np.float32([0.0]))[0] == 0x80000000  # -0 counts as +0
    back = keys(L, 5, x).view(np.float32)
    assert np.array_equal(back[x != 0], x[x != 0]) and (back[x == 0] == 0).all() and not np.signbit(back[x == 0]).any()
    assert np.array_equal(keys(L, 6, x[:5000]).astype(np.int64), np.arange(5000))  # the row comes back
    # larger key = better: higher score first, and among equal scores the SMALLER row
    assert L.gh_test_sparse_key(2.0, 7) > L.gh_test_sparse_key(1.0, 3)
    assert L.gh_test_sparse_key(1.0, 3) > L.gh_test_sparse_key(1.0, 4)
    assert L.gh_test_sparse_key(0.0, 5) == L.gh_test_sparse_key(-0.0, 5)
    assert L.gh_test_sparse_key(0.0, 5) > L.gh_test_sparse_key(-1e-30, 0)


def test_written_counts_what_the_reference_returns(L):
    """xvec.go:379-446: every admissible document is ranked (score descending), the list is cut to k, then zero scores are
    dropped.  Brute force over small populations."""
    for pos, neg, zeros, k in itertools.product(range(0, 7), range(0, 7), range(0, 7), (1, 2, 3, 5, 8, 30)):
        scores = [1.0] * pos + [0.0] * zeros + [-1.0] * neg
        top = sorted(scores, reverse=True)[:k]
        expect = sum(1 for s in top if s != 0.0)
        assert L.gh_test_sparse_written(pos, neg, pos + neg + zeros, k) == expect, (pos, neg, zeros, k)
