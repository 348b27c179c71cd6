"""GPU parity: the SYMMETRIC form of the sparse all-pairs pass (csrc/sparse_kernels.hpp, SymArgs; round 6).  When every stored row is a
query and leaves out only itself, a pair of whole-query rows is walked once, by the shorter row, and the score is delivered to the other
row's ranking (published bounds, foreign lists, a merge, and the unsymmetric walk again for a row whose list overflowed).  Rows, score
bits, counts and padding must equal the unsymmetric walk's -- which tests/test_gpu_vectors_sparse.py holds against the oracle -- and the
oracle's (storage/vectors/xvec.go:379-446 through oracle/)."""
import numpy as np
import pytest

from gorse_amd import capi, synth
from sparse_cases import check_against_oracle as check, random_csr, rows_of

pytestmark = pytest.mark.gpu


@pytest.fixture
def hooks():
    L = capi.lib()
    yield L
    L.gorse_hip_test_set_sparse_sym(-1, 0, 0, 0)
    L.gorse_hip_test_set_sparse_front(1)
    L.gorse_hip_test_set_sparse_slots(0)
    L.gorse_hip_test_set_sparse_head(-1)
    L.gorse_hip_test_set_sparse_tile(0)
    L.gorse_hip_test_set_sparse_split(2048)
    L.gorse_hip_test_set_sparse_heavy(16384)
    L.gorse_hip_test_set_sparse_atomic(-1)


def _bits(x):
    return x.view(np.uint32) if x.dtype == np.float32 else x


def _same(x, y):
    for a, b in zip(x, y):
        assert np.array_equal(_bits(a), _bits(b))


def _both(hooks, s, k, caps=(0, 0, 0)):
    """(unsymmetric, symmetric) results of all_pairs(k) and the symmetric call's statistics"""
    hooks.gorse_hip_test_set_sparse_sym(0, 0, 0, 0)
    plain = s.all_pairs(k)
    assert s.sym_stats()[0] == 0
    hooks.gorse_hip_test_set_sparse_sym(1, *caps)
    sym = s.all_pairs(k)
    st = s.sym_stats()
    hooks.gorse_hip_test_set_sparse_sym(-1, 0, 0, 0)
    return plain, sym, st


@pytest.mark.parametrize("k", [1, 7, 100, 130, 300, 1024])
@pytest.mark.parametrize("shape", ["signed_zipf", "positive", "short_rows"])
def test_symmetric_pass_equals_the_unsymmetric_one(oracle, hooks, shape, k):
    rng = np.random.default_rng(600 + k)
    if shape == "signed_zipf":  # long and heavy rows in front (their own kernels), negative and cancelling scores
        ptr, idx, val = random_csr(rng, 5000, 90, 0, 14, neg=True, zipf=True)
        hooks.gorse_hip_test_set_sparse_tile(256)
        hooks.gorse_hip_test_set_sparse_split(6)
        hooks.gorse_hip_test_set_sparse_heavy(9)
    elif shape == "positive":  # the shape of the IDF collections: everything positive, most pairs share something
        ptr, idx, val = random_csr(rng, 3000, 300, 1, 40, zipf=True)
        hooks.gorse_hip_test_set_sparse_tile(512)
    else:  # most rows have fewer than k partners: no bound is ever published, everything delivered is appended
        ptr, idx, val = random_csr(rng, 4000, 6000, 0, 4)
    s = capi.Sparse(ptr, idx, val)
    plain, sym, st = _both(hooks, s, k)
    assert st[0] == 1
    _same(plain, sym)
    sample = list(range(0, ptr.size - 1, 97))
    check(oracle, ptr, idx, val, k, [x[sample] for x in sym], rows_of(ptr, idx, val, sample), sample)
    s.close()


def test_overflowing_foreign_lists_send_rows_through_the_unsymmetric_walk(oracle, hooks):
    rng = np.random.default_rng(77)
    ptr, idx, val = random_csr(rng, 6000, 200, 0, 12, neg=True, zipf=True)
    hooks.gorse_hip_test_set_sparse_tile(256)
    hooks.gorse_hip_test_set_sparse_split(10)
    s = capi.Sparse(ptr, idx, val)
    for caps in ((1, 1, 1), (3, 2, 1), (64, 8, 2)):
        plain, sym, st = _both(hooks, s, 20, caps)
        assert st[0] == 1 and st[1] > 0, st  # rows were redone
        assert st[3] > caps[2]
        _same(plain, sym)
    sample = list(range(0, 6000, 61))
    check(oracle, ptr, idx, val, 20, [x[sample] for x in sym], rows_of(ptr, idx, val, sample), sample)
    s.close()


def test_equal_rows_and_equal_scores(oracle, hooks):
    """blocks of identical rows: equal lengths (the order of the scratch ids decides who walks a pair) and equal scores (ascending row
    among them, whether a key came from the own walk or was delivered)"""
    rng = np.random.default_rng(5)
    base_ptr, base_idx, base_val = random_csr(rng, 40, 30, 1, 8)
    reps = rng.integers(0, 40, 1500)
    ptr = [0]
    idx, val = [], []
    for r in reps:
        idx.append(base_idx[base_ptr[r]:base_ptr[r + 1]])
        val.append(base_val[base_ptr[r]:base_ptr[r + 1]])
        ptr.append(ptr[-1] + idx[-1].size)
    ptr, idx, val = np.array(ptr, np.int64), np.concatenate(idx), np.concatenate(val)
    hooks.gorse_hip_test_set_sparse_tile(256)
    s = capi.Sparse(ptr, idx, val)
    for k in (5, 64, 200):
        plain, sym, st = _both(hooks, s, k)
        assert st[0] == 1
        _same(plain, sym)
        check(oracle, ptr, idx, val, k, [x[:200] for x in sym], rows_of(ptr, idx, val, range(200)), list(range(200)))
    s.close()


def test_calls_that_are_not_all_pairs_keep_the_unsymmetric_walk(hooks):
    rng = np.random.default_rng(9)
    ptr, idx, val = random_csr(rng, 800, 100, 0, 10)
    s = capi.Sparse(ptr, idx, val)
    s.all_pairs(5)
    assert s.sym_stats()[0] == 0  # one row group: nothing to leave out, the default keeps the unsymmetric walk
    hooks.gorse_hip_test_set_sparse_tile(256)
    s.close()
    s = capi.Sparse(ptr, idx, val)
    s.all_pairs(5)
    assert s.sym_stats()[0] == 1
    s.all_pairs(5, 0, 799)
    assert s.sym_stats()[0] == 0
    s.all_pairs(5, exclude_self=False)
    assert s.sym_stats()[0] == 0
    s.set_mask((rng.random(800) < 0.5).astype(np.uint8))
    s.all_pairs(5)
    assert s.sym_stats()[0] == 0
    s.set_mask(None)
    qp, qi, qv = random_csr(rng, 800, 100, 0, 10)
    s.search(qp, qi, qv, 5)
    assert s.sym_stats()[0] == 0
    s.close()


def test_idf_collection_of_a_synthetic_shard(oracle, hooks):
    """the writer's own shape (logics/vector_writer.go:192-209: IDF weights over the users of an item), default switches: both forms,
    the statistics of the symmetric one (fewer postings walked, no row redone), a sample against the oracle"""
    data = synth.s_ml1m()
    ptr, idx, val = synth.idf_vectors(data.iptr, data.iidx, data.U)
    s = capi.Sparse(ptr, idx, val)
    hooks.gorse_hip_test_set_sparse_sym(0, 0, 0, 0)
    plain = s.all_pairs(100)
    walked_plain = s.last_stats()[0]
    hooks.gorse_hip_test_set_sparse_sym(1, 0, 0, 0)
    sym = s.all_pairs(100)
    walked_sym = s.last_stats()[0]
    st = s.sym_stats()
    assert st[0] == 1 and st[1] == 0
    assert walked_sym < walked_plain
    _same(plain, sym)
    sample = list(range(0, ptr.size - 1, max(1, (ptr.size - 1) // 64)))
    check(oracle, ptr, idx, val, 100, [x[sample] for x in sym], rows_of(ptr, idx, val, sample), sample)
    s.close()


@pytest.mark.parametrize("k", [3, 100, 300])
@pytest.mark.parametrize("split", [38, 33, 20])
def test_front_group_of_the_long_rows(oracle, hooks, split, k):
    """The rows longer than the split threshold at creation get a row group of their own (phantom scratch ids behind them) when they are
    fewer than a group holds (split 38: ~150 of 3000 rows, 33: ~500, 20: more than the 512 of a group -> plain numbering): every kind of
    call -- all pairs in both forms, a query range, searches with masks and exclusions -- against the plain numbering and the oracle."""
    rng = np.random.default_rng(800 + split)
    ptr, idx, val = random_csr(rng, 3000, 300, 1, 40, neg=(split == 33), zipf=True)
    n_long = int((np.diff(ptr) > split).sum())
    assert (n_long < 512) == (split != 20) and n_long > 0
    hooks.gorse_hip_test_set_sparse_tile(512)
    hooks.gorse_hip_test_set_sparse_split(split)
    hooks.gorse_hip_test_set_sparse_heavy(39)  # the longest rows: the dense-vector kernel
    hooks.gorse_hip_test_set_sparse_front(0)
    plain_index = capi.Sparse(ptr, idx, val)
    hooks.gorse_hip_test_set_sparse_front(1)
    s = capi.Sparse(ptr, idx, val)
    ref, sym, st = _both(hooks, plain_index, k)
    unsym2, sym2, st2 = _both(hooks, s, k)
    assert st[0] == 1 and st2[0] == 1
    _same(ref, sym)
    _same(ref, unsym2)
    _same(ref, sym2)  # (fewer long rows than a group holds: the front delivered)
    hooks.gorse_hip_test_set_sparse_sym(2, 0, 0, 0)  # the front does not deliver
    _same(ref, s.all_pairs(k))
    assert s.sym_stats()[0] == 1
    hooks.gorse_hip_test_set_sparse_sym(1, 1, 1, 1)  # every foreign list overflows
    _same(ref, s.all_pairs(k))
    assert s.sym_stats()[1] > 0
    hooks.gorse_hip_test_set_sparse_sym(-1, 0, 0, 0)
    sample = list(range(0, 3000, 41))
    check(oracle, ptr, idx, val, k, [x[sample] for x in sym2], rows_of(ptr, idx, val, sample), sample)
    _same(plain_index.all_pairs(k, 100, 900, exclude_self=False), s.all_pairs(k, 100, 900, exclude_self=False))
    mask = (rng.random(3000) < 0.7).astype(np.uint8)
    qp, qi, qv = random_csr(rng, 40, 320, 0, 60, neg=True)
    excl = rng.integers(-1, 3000, 40).astype(np.int64)
    for index in (plain_index, s):
        index.set_mask(mask)
    got = s.search(qp, qi, qv, k, exclude=excl)
    _same(plain_index.search(qp, qi, qv, k, exclude=excl), got)
    check(oracle, ptr, idx, val, k, got, rows_of(qp, qi, qv, range(40)), list(excl), mask)
    _same(plain_index.all_pairs(k), s.all_pairs(k))
    plain_index.close()
    s.close()


@pytest.mark.parametrize("k", [5, 100])
@pytest.mark.parametrize("head", [-1, 0, 1, 1000])
def test_front_cut_above_the_split_threshold(oracle, hooks, head, k):
    """The front's rows are the ones longer than a cut of their own (here 36 of at most 40 entries, the split threshold of every other call
    20): a symmetric pass splits only them into per-group work items, the rows between the two thresholds walk as single items.  With 0, 1 and
    all head groups (the groups a whole-query item visits one by one; the super-visits behind them must start behind the front too)."""
    rng = np.random.default_rng(900)
    ptr, idx, val = random_csr(rng, 4000, 300, 1, 40, zipf=True)
    assert 100 < int((np.diff(ptr) > 36).sum()) < 512
    hooks.gorse_hip_test_set_sparse_tile(512)
    hooks.gorse_hip_test_set_sparse_split(20)
    hooks.gorse_hip_test_set_sparse_heavy(39)
    hooks.gorse_hip_test_set_sparse_head(head)
    hooks.gorse_hip_test_set_sparse_front(36)
    s = capi.Sparse(ptr, idx, val)
    plain, sym, st = _both(hooks, s, k)
    assert st[0] == 1
    _same(plain, sym)
    sample = list(range(0, 4000, 53))
    check(oracle, ptr, idx, val, k, [x[sample] for x in sym], rows_of(ptr, idx, val, sample), sample)
    got = s.all_pairs(k, 17, 3000)  # a range: the unsymmetric walk with its own split threshold, on the front's numbering
    check(oracle, ptr, idx, val, k, [x[:60] for x in got], rows_of(ptr, idx, val, range(17, 77)), list(range(17, 77)))
    s.close()


def test_random_configurations_of_the_symmetric_pass(oracle, hooks):
    """Thirty random small collections (empty rows, equal rows, signed values, fewer rows than k, one row group or dozens) under random settings
    of every switch the pass has (rows per group, split / heavy thresholds, the front's cut, head groups, workgroups, accumulation form, foreign
    list capacities, with and without the front's delivery): the symmetric pass equals the unsymmetric one everywhere and the oracle on a sample."""
    rng = np.random.default_rng(4242)
    for case in range(30):
        rows, dims = int(rng.integers(2, 2000)), int(rng.integers(1, 200))
        hi = int(rng.integers(0, min(dims, 40) + 1))
        ptr, idx, val = random_csr(rng, rows, dims, 0, hi, neg=bool(rng.integers(0, 2)), zipf=bool(rng.integers(0, 2)))
        if case % 5 == 0:  # blocks of identical rows
            keep = rng.integers(0, rows, rows)
            lens = np.diff(ptr)[keep]
            idx = np.concatenate([idx[ptr[r]:ptr[r + 1]] for r in keep]) if lens.sum() else idx[:0]
            val = np.concatenate([val[ptr[r]:ptr[r + 1]] for r in keep]) if lens.sum() else val[:0]
            ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        hooks.gorse_hip_test_set_sparse_tile(int(rng.choice([256, 512, 2048])))
        hooks.gorse_hip_test_set_sparse_split(int(rng.choice([0, 1, 3, 8, 20, 2048])))
        hooks.gorse_hip_test_set_sparse_heavy(int(rng.choice([0, 2, 12, 30, 16384])))
        hooks.gorse_hip_test_set_sparse_head(int(rng.choice([-1, 0, 1, 2, 1000])))
        hooks.gorse_hip_test_set_sparse_slots(int(rng.choice([0, 3, 64])))
        hooks.gorse_hip_test_set_sparse_atomic(int(rng.choice([-1, 0, 1])))
        hooks.gorse_hip_test_set_sparse_front(int(rng.choice([0, 1, 1, 5, 15, 30])))
        s = capi.Sparse(ptr, idx, val)
        k = int(rng.choice([1, 2, 7, 64, 100, 129, 300, 1024]))
        caps = (0, 0, 0) if rng.random() < 0.6 else tuple(int(x) for x in rng.choice([1, 4, 64], 3))
        hooks.gorse_hip_test_set_sparse_sym(0, 0, 0, 0)
        plain = s.all_pairs(k)
        hooks.gorse_hip_test_set_sparse_sym(int(rng.choice([1, 1, 2])), *caps)
        sym = s.all_pairs(k)
        hooks.gorse_hip_test_set_sparse_sym(-1, 0, 0, 0)
        for a, b in zip(plain, sym):
            assert np.array_equal(_bits(a), _bits(b)), case
        sample = sorted(set(int(x) for x in rng.integers(0, rows, 12)))
        check(oracle, ptr, idx, val, k, [x[sample] for x in sym], rows_of(ptr, idx, val, sample), sample)
        s.close()
