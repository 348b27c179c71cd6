"""The tile schedule of the symmetric all-pairs sweep (gorse_amd/csrc/topk_sym.hpp, included by topk_sweep_kernel<..., SYM> and by
the host library's hook): whatever the shape -- a query range that starts inside the index, ends inside it, ends inside a tile, a
last query block that is partial -- every (query, row) pair is scored exactly once, either by the query's own workgroup or along
the rows of a transposed tile of a later block's workgroup, and the workgroups multiply about half the tiles of the square sweep
when the queries are all the rows.  Reference semantics: ann.Bruteforce scores every vector for every query
(common/ann/bruteforce.go:39-83)."""
import ctypes

import numpy as np
import pytest

from gorse_amd import cf


def cover(n, q0, nq, tile_rows, bq):
    lib = ctypes.CDLL(cf.HOST_LIB)
    lib.gh_test_topk_sym_cover.restype = ctypes.c_int64
    lib.gh_test_topk_sym_cover.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_void_p]
    c = np.zeros((nq, n), np.int32)
    tiles = lib.gh_test_topk_sym_cover(n, q0, nq, tile_rows, bq, c.ctypes.data)
    return c, tiles


@pytest.mark.parametrize("n,q0,nq", [(96, 0, 96), (100, 0, 100), (97, 0, 97), (200, 0, 64), (200, 64, 136), (200, 64, 70),
                                     (203, 8, 150), (64, 0, 33), (500, 120, 200), (333, 0, 333)])
@pytest.mark.parametrize("tile_rows,bq", [(8, 32), (8, 8), (4, 16)])
def test_every_pair_is_scored_exactly_once(n, q0, nq, tile_rows, bq):
    if q0 % tile_rows:
        pytest.skip("the query range starts on a tile boundary (the library falls back to the square sweep otherwise)")
    c, tiles = cover(n, q0, nq, tile_rows, bq)
    assert (c == 1).all(), np.argwhere(c != 1)[:5]
    all_tiles = -(-n // tile_rows)
    blocks = -(-nq // bq)
    assert tiles <= blocks * all_tiles
    if q0 == 0 and nq == n and blocks >= 4:
        assert tiles <= 0.5 * blocks * all_tiles * (1 + 2.0 / blocks) + blocks  # the triangle, not the square


def test_the_kernels_own_shape():
    # C4's geometry scaled down 128-fold: 128-row tiles, 512-query blocks
    c, tiles = cover(8000, 0, 8000, 128, 512)
    assert (c == 1).all()
    c, tiles = cover(8000, 1024, 5000, 128, 512)
    assert (c == 1).all()
