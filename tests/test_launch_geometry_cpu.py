"""The geometry of the BPR chunk preparation by user bins (gorse_amd/csrc/bpr_bins.hpp, included by csrc/bpr.hip's launch code and by
the host library's hooks): for every shape the product can meet, the bins fit the kernels' LDS, the tile x bin matrix a handle
allocates holds every chunk up to its capacity, and the four passes -- restated on the CPU in the kernels' arithmetic -- leave every
sample in the run of its user with the run offsets the update kernel reads.  Reference semantics: the samples of an epoch
(model/cf/model.go:449-468) may be applied in any order (common/parallel/parallel.go:44-68); the preparation only groups them."""
import ctypes as C

import numpy as np
import pytest

from gorse_amd import cf

K_MAX_BINS, K_MAX_SHIFT = 8192, 11


def lib():
    L = C.CDLL(cf.HOST_LIB)
    L.gh_test_bpr_bins_geometry.argtypes = [C.c_int64, C.c_int64, C.c_void_p]
    L.gh_test_bpr_bins_matrix_words.restype = C.c_int64
    L.gh_test_bpr_bins_matrix_words.argtypes = [C.c_int64, C.c_int64]
    L.gh_test_bpr_bins_emulate.restype = C.c_int32
    L.gh_test_bpr_bins_emulate.argtypes = [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def geometry(U, n):
    out = np.zeros(5, np.int64)
    lib().gh_test_bpr_bins_geometry(U, n, out.ctypes.data)
    return dict(shift=int(out[0]), nbins=int(out[1]), tile=int(out[2]), tiles=int(out[3]), ok=bool(out[4]))


def test_geometry_fits_the_kernels_for_every_shape():
    users = [1, 2, 59, 60, 943, 4096, 6040, 125_000, 1_000_000, 10_000_000, 16_000_000, (1 << 24) - 1, 1 << 24, 40_000_000, (1 << 31) - 1]
    samples = [1, 2, 4095, 4096, 4097, 99_057, 994_169, 4_194_304, 12_500_000, 33_554_432, 110_000_000, 1 << 27]
    for U in users:
        for n in samples:
            g = geometry(U, n)
            assert g["nbins"] == (U >> g["shift"]) + 1 and 1 <= g["nbins"] <= K_MAX_BINS - 1  # one LDS counter per bin, bin_start[nbins] in reach
            assert g["ok"] == (g["shift"] <= K_MAX_SHIFT)                                   # ... and per user id of a bin
            assert g["ok"] or U >= (K_MAX_BINS - 1) << K_MAX_SHIFT                          # only beyond ~16M users
            assert g["tile"] in (4096, 8192, 16384, 32768, 65536) and g["tiles"] == -(-n // g["tile"])
            assert g["tiles"] <= max(512, -(-n // 65536))
            assert (U >> g["shift"]) < g["nbins"]                                           # the bin of key U (no user) exists
            # a chunk's matrix fits what a handle of ANY capacity >= n allocated
            for cap in (n, n + 1, 2 * n, 1 << 27):
                if cap >= n:
                    assert g["tiles"] * g["nbins"] <= lib().gh_test_bpr_bins_matrix_words(U, cap), (U, n, cap, g)


@pytest.mark.parametrize("U,n,empty_every", [(60, 20_000, 3), (5000, 300_000, 0), (300_000, 700_000, 3), (7, 9000, 0), (70_000, 65_537, 5)])
def test_the_four_passes_group_every_sample_by_its_user(U, n, empty_every):
    rng = np.random.default_rng(U + n)
    key = rng.integers(0, U, n).astype(np.int32)
    if empty_every:
        key[rng.integers(0, n, n // 50)] = -1  # samples whose user draw failed: sorted behind the last user
    bucket, ps, pu = np.zeros(U + 2, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    assert lib().gh_test_bpr_bins_emulate(U, key.ctypes.data, n, bucket.ctypes.data, ps.ctypes.data, pu.ctypes.data) == 1
    k = np.where(key < 0, U, key)
    assert bucket[0] == 0 and bucket[U + 1] == n and np.array_equal(np.diff(bucket), np.bincount(k, minlength=U + 1))
    assert np.array_equal(np.sort(ps), np.arange(n)) and np.array_equal(pu, key[ps])         # a permutation, each with its own key
    run_of = np.repeat(np.arange(U + 1), np.diff(bucket))
    assert np.array_equal(run_of, np.where(pu < 0, U, pu))                                   # every pair inside the run of its user


def test_als_long_row_threshold_follows_the_side():
    """csrc/als_plan.hpp (included by als_build_plan): a power of two in [256 (512 from nFactors 64 on), 4096], non-decreasing in the
    side's size, 4096 for the sides of C5 (50M entries) and 256 / 512 for S-ml1m's (model/cf/model_test.go:93-104: the shape of
    TestCCD_MovieLens)."""
    L = C.CDLL(cf.HOST_LIB)
    L.gh_test_als_long_row.restype = C.c_int64
    L.gh_test_als_long_row.argtypes = [C.c_int64, C.c_int32]
    for d in (8, 16, 48, 64, 128):
        floor = 512 if d >= 64 else 256
        prev = 0
        for entries in [0, 1, 1000, 994_169, 1 << 20, 3_000_000, 10_000_000, 16_777_216, 50_000_000, 1_000_000_000]:
            t = L.gh_test_als_long_row(entries, d)
            assert floor <= t <= 4096 and t & (t - 1) == 0 and t >= prev
            assert t == floor or t // 2 < entries / 4096 or t == 4096
            prev = t
        assert L.gh_test_als_long_row(994_169, d) == floor and L.gh_test_als_long_row(50_000_000, d) == 4096
