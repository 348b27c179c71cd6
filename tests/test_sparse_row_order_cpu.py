"""The scratch numbering of a sparse index's rows (gorse_amd/csrc/sparse_host.hpp -- the header gorse_sparse_create includes), checked on the
CPU through the host library's hook gh_test_sparse_row_order: longest row first, ties by row; the rows longer than the split threshold get a
row group of their own -- phantom ids behind them -- exactly when they are fewer than a group holds (the front)."""
import ctypes as C

import numpy as np
import pytest

from gorse_amd import cf


@pytest.fixture(scope="module")
def L():
    lib = C.CDLL(cf.HOST_LIB)
    lib.gh_test_sparse_row_order.argtypes = [C.c_int64, C.POINTER(C.c_int64), C.c_int64, C.c_int64, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.gh_test_sparse_row_order.restype = None
    return lib


def order(L, lens, cut, group):
    n = len(lens)
    ptr = np.concatenate([[7], 7 + np.cumsum(lens)]).astype(np.int64)  # (a CSR that does not start at 0)
    new_of, orig_of, out = np.empty(n, np.int32), np.full(n + max(group, 0), -7, np.int32), np.zeros(3, np.int64)
    L.gh_test_sparse_row_order(n, ptr.ctypes.data_as(C.POINTER(C.c_int64)), cut, group, new_of.ctypes.data_as(C.POINTER(C.c_int32)),
                               orig_of.ctypes.data_as(C.POINTER(C.c_int32)), out.ctypes.data_as(C.POINTER(C.c_int64)))
    return new_of, orig_of[:out[0]], int(out[0]), int(out[1]), int(out[2])


@pytest.mark.parametrize("seed", range(6))
def test_row_order_with_and_without_a_front(L, seed):
    rng = np.random.default_rng(seed)
    n, group = int(rng.integers(1, 700)), int(rng.choice([16, 64, 256]))
    lens = rng.integers(0, 40, n)
    by_len = np.argsort(-lens, kind="stable")
    for cut in (0, 5, 20, 35, 38, 39, 100):
        new_of, orig_of, Np, n_front, pad = order(L, lens, cut, group)
        n_long = int((lens > cut).sum()) if cut > 0 else 0
        if 0 < n_long < group and n_long < n:
            assert (n_front, pad, Np) == (n_long, group - n_long, n + group - n_long)
            assert (orig_of[n_front:group] == -1).all()  # the phantom ids
            assert np.array_equal(orig_of[:n_front], by_len[:n_front]) and np.array_equal(orig_of[group:], by_len[n_front:])
            assert (lens[orig_of[:n_front]] > cut).all() and (lens[orig_of[group:]] <= cut).all()  # group 0 = the long rows, alone
        else:
            assert (n_front, pad, Np) == (0, 0, n)
            assert np.array_equal(orig_of, by_len)
        real = orig_of >= 0
        assert real.sum() == n and np.array_equal(np.sort(orig_of[real]), np.arange(n))
        assert np.array_equal(orig_of[new_of], np.arange(n))  # the two maps agree
        ranked = lens[orig_of[real]]
        assert (np.diff(ranked) <= 0).all()  # longest first through the phantoms


def test_edges(L):
    assert order(L, [3], 1, 64)[2:] == (1, 0, 0)            # the only row is long: nothing behind a front
    assert order(L, [3, 1], 1, 64)[2:] == (2 + 63, 1, 63)   # one long row, one behind it
    assert order(L, [3, 3, 1], 1, 2)[2:] == (3, 0, 0)       # as many long rows as a group holds: plain numbering
    assert order(L, [0, 0, 0], 5, 64)[2:] == (3, 0, 0)      # no long row
    new_of, orig_of, Np, n_front, pad = order(L, [2, 9, 2, 9], 5, 4)
    assert orig_of.tolist() == [1, 3, -1, -1, 0, 2] and new_of.tolist() == [4, 0, 5, 1]  # ties by row
