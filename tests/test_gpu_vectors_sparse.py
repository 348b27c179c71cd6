"""GPU parity: exact top-k over sparse vectors (the sparse collections of vectors.Database: storage/vectors/
database.go:90-97, xvec.go:241-247; filled by the IDF writers of logics/vector_writer.go:192-209) through the C ABI
(gorse_sparse_*), against the oracle's merge-order sparse dot: rows, score bits, counts and padding."""
import numpy as np
import pytest

import vectors_suite as S
from gorse_amd import capi, synth
from gorse_amd import vectors as V
from sparse_cases import TIE_EXPECT, check_against_oracle as check, random_csr, rows_of, tie_case

pytestmark = pytest.mark.gpu


def test_all_pairs_equals_oracle(oracle):
    rng = np.random.default_rng(100)
    ptr, idx, val = random_csr(rng, 600, 400, 0, 30)
    s = capi.Sparse(ptr, idx, val)
    capi.lib().gorse_hip_test_set_sparse_sym(1, 0, 0, 0)  # (one row group: the default would keep the unsymmetric walk)
    try:
        for k in (7, 64, 100):
            got = s.all_pairs(k)
            check(oracle, ptr, idx, val, k, got, rows_of(ptr, idx, val, range(600)), list(range(600)))
        # statistics of the last call: postings walked = posting-list lengths summed over the queries' indices in the unsymmetric
        # walk (the symmetric form of an all-pairs pass, tests/test_gpu_sparse_sym.py, walks fewer where there is a group to leave out)
        lens = np.bincount(idx, minlength=400)
        postings, hits = s.last_stats()
        assert s.sym_stats()[0] == 1 and postings <= int(lens[idx].sum())
        capi.lib().gorse_hip_test_set_sparse_sym(0, 0, 0, 0)
        _same(got, s.all_pairs(100))
        postings, hits = s.last_stats()
        assert postings == int(lens[idx].sum())
    finally:
        capi.lib().gorse_hip_test_set_sparse_sym(-1, 0, 0, 0)
    got = s.all_pairs(5, q_begin=590, q_end=597, exclude_self=False)
    check(oracle, ptr, idx, val, 5, got, rows_of(ptr, idx, val, range(590, 597)), [-1] * 7)


@pytest.mark.parametrize("k", [3, 10, 64, 65, 200, 513, 1024])
def test_many_hits_overflow_the_ranking_buffer(oracle, k):
    """far more hits than the 2*KP slots of the LDS buffer, negative scores included"""
    rng = np.random.default_rng(7)
    ptr, idx, val = random_csr(rng, 5000, 30, 3, 12, neg=True)
    qp, qi, qv = random_csr(rng, 9, 30, 8, 20, neg=True)
    s = capi.Sparse(ptr, idx, val)
    got = s.search(qp, qi, qv, k)
    check(oracle, ptr, idx, val, k, got, rows_of(qp, qi, qv, range(9)), [-1] * 9)


def test_ties_zero_scores_mask_and_exclude(oracle):
    """equal scores rank by ascending row; zero scores are dropped AFTER the cut to k (xvec.go:419-421: they use up
    slots, so negative scores appear only when every zero-score row fits too); mask / exclude"""
    ptr, idx, val, (qp, qi, qv), mask, excl = tie_case()
    s = capi.Sparse(ptr, idx, val)
    s.set_mask(mask)
    for k, expect in TIE_EXPECT.items():
        got = s.search(qp, qi, qv, k, exclude=excl)
        check(oracle, ptr, idx, val, k, got, rows_of(qp, qi, qv, range(3)), list(excl), mask)
        assert list(got[0][0, :got[2][0]]) == expect
    s.set_mask(None)
    got = s.search(qp, qi, qv, 20, exclude=excl)
    check(oracle, ptr, idx, val, 20, got, rows_of(qp, qi, qv, range(3)), list(excl), None)


def test_test_sparse_of_the_reference():
    """storage/vectors/database_test.go:198-224 (TestSparse): the query {1: 1, 100: 2} finds "match" (1*1 + 2*2 = 5) then
    "old" (1*1 + 2*1 = 3); "other" shares no index and is not returned"""
    ptr = np.array([0, 2, 4, 6], np.int64)
    idx = np.array([1, 100, 1, 100, 2, 200], np.uint32)
    val = np.array([1, 1, 1, 2, 1, 2], np.float32)
    s = capi.Sparse(ptr, idx, val)
    i, sc, cnt = s.search(np.array([0, 2], np.int64), np.array([1, 100], np.uint32), np.array([1, 2], np.float32), 10)
    assert cnt[0] == 2 and list(i[0, :2]) == [1, 0] and list(sc[0, :2]) == [5.0, 3.0]


def test_scratch_reuse_across_calls(oracle):
    """a workgroup's LDS accumulators serve one query after the other (every read-back leaves them zero); calls repeat"""
    rng = np.random.default_rng(11)
    ptr, idx, val = random_csr(rng, 3000, 200, 1, 8, zipf=True)
    s = capi.Sparse(ptr, idx, val)
    first = s.all_pairs(9)
    qp, qi, qv = random_csr(rng, 40, 260, 1, 12)
    mid = s.search(qp, qi, qv, 30)
    check(oracle, ptr, idx, val, 30, mid, rows_of(qp, qi, qv, range(40)), [-1] * 40)
    again = s.all_pairs(9)
    for a, b in zip(first, again):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
    sample = list(range(0, 3000, 97))
    check(oracle, ptr, idx, val, 9, [x[sample] for x in again], rows_of(ptr, idx, val, sample), sample)


def test_users_item_to_item_ml100k_shape(oracle):
    """the "users" item-to-item vectors (item -> its users ascending, value sqrt(idf(user))) of an S-ml100k-shaped
    dataset: sampled rows against the oracle, every row through the properties that need no oracle"""
    data = synth.s_ml100k()
    ptr, idx, val = synth.idf_vectors(data.iptr, data.iidx, data.U)
    k = 100
    s = capi.Sparse(ptr, idx, val)
    s.set_profiling(True)
    out_idx, out_sc, out_cnt = s.all_pairs(k)
    launches, ms = s.get_profile()
    assert launches == 1 and ms > 0
    sample = list(range(0, data.I, 41))
    check(oracle, ptr, idx, val, k, (out_idx[sample], out_sc[sample], out_cnt[sample]), rows_of(ptr, idx, val, sample), sample)
    properties(ptr, out_idx, out_sc, out_cnt, k)


def properties(ptr, out_idx, out_sc, out_cnt, k):
    N = out_idx.shape[0]
    valid = np.arange(k)[None, :] < out_cnt[:, None]
    assert (out_idx[valid] >= 0).all() and (out_idx[~valid] == -1).all() and np.isneginf(out_sc[~valid]).all()
    assert not (out_idx == np.arange(N)[:, None]).any()  # exclude_self
    # descending scores, equal scores in ascending row order
    a, b = out_sc[:, :-1], out_sc[:, 1:]
    both = valid[:, 1:]
    assert (a[both] >= b[both]).all()
    eq = both & (a == b)
    assert (out_idx[:, :-1][eq] < out_idx[:, 1:][eq]).all()
    # the sparse dot is symmetric bit for bit (same common indices, same order, commutative products): whenever j is
    # in i's list and i in j's, the two scores agree
    pos = {}
    for i in range(N):
        for t in range(out_cnt[i]):
            pos[(i, int(out_idx[i, t]))] = out_sc[i, t]
    checked = 0
    for (i, j), sc in pos.items():
        other = pos.get((j, i))
        if other is not None:
            assert np.float32(sc).view(np.uint32) == np.float32(other).view(np.uint32)
            checked += 1
    assert checked > N
    # rows without entries have no hits
    empty = np.diff(ptr) == 0
    assert (out_cnt[empty] == 0).all()


def _same(x, y):
    for a, b in zip(x, y):
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)


@pytest.fixture
def hooks():
    L = capi.lib()
    yield L
    L.gorse_hip_test_set_sparse_slots(0)
    L.gorse_hip_test_set_sparse_head(-1)
    L.gorse_hip_test_set_sparse_tile(0)
    L.gorse_hip_test_set_sparse_split(2048)
    L.gorse_hip_test_set_sparse_heavy(16384)
    L.gorse_hip_test_set_sparse_atomic(-1)


@pytest.mark.parametrize("k", [5, 70, 300])
def test_long_queries_are_split_over_the_row_groups(oracle, k, hooks):
    """queries with more entries than the threshold (2048 by default, 6 here) are answered by one work item per row group
    (36 groups of 256 rows here), each with its own ranking, and a merge: same results, in one call together with unsplit
    queries, masks, exclusions, negative and cancelling scores"""
    rng = np.random.default_rng(47)
    ptr, idx, val = random_csr(rng, 9000, 80, 0, 14, neg=True, zipf=True)
    n_long = int((np.diff(ptr) > 6).sum())
    assert n_long > 100 and n_long < 8000
    mask = (rng.random(9000) < 0.8).astype(np.uint8)
    hooks.gorse_hip_test_set_sparse_tile(256)
    s = capi.Sparse(ptr, idx, val)
    hooks.gorse_hip_test_set_sparse_split(6)
    hooks.gorse_hip_test_set_sparse_heavy(9)  # more than 9 entries: scored row by row against a dense copy of the query
    assert int((np.diff(ptr) > 9).sum()) > 50
    sample = list(range(0, 9000, 23))
    got = s.all_pairs(k)
    check(oracle, ptr, idx, val, k, [x[sample] for x in got], rows_of(ptr, idx, val, sample), sample)
    s.set_mask(mask)
    qp, qi, qv = random_csr(rng, 30, 95, 0, 25, neg=True)
    excl = rng.integers(-1, 9000, 30).astype(np.int64)
    split = s.search(qp, qi, qv, k, exclude=excl)
    check(oracle, ptr, idx, val, k, split, rows_of(qp, qi, qv, range(30)), list(excl), mask)
    hooks.gorse_hip_test_set_sparse_split(0)
    _same(split, s.search(qp, qi, qv, k, exclude=excl))  # the same call, one work item per query
    s.set_mask(None)
    _same(got, s.all_pairs(k))


@pytest.mark.parametrize("atomic", [0, 1])
def test_both_accumulation_forms(oracle, atomic, hooks):
    """ds_add_f32 (fire and forget, applied in issue order) and the load / add / store form give the oracle's bits: rows hit
    by many posting lists of one query (a popularity law over few indices), dense and sparse tiles, cancelling terms"""
    rng = np.random.default_rng(5 + atomic)
    ptr, idx, val = random_csr(rng, 20000, 300, 1, 40, neg=True, zipf=True)
    hooks.gorse_hip_test_set_sparse_atomic(atomic)
    hooks.gorse_hip_test_set_sparse_tile(1024)
    hooks.gorse_hip_test_set_sparse_split(200)
    s = capi.Sparse(ptr, idx, val)
    sample = list(range(0, 20000, 397))
    got = s.all_pairs(50)
    check(oracle, ptr, idx, val, 50, [x[sample] for x in got], rows_of(ptr, idx, val, sample), sample)
    qp, qi, qv = random_csr(rng, 12, 300, 100, 280, neg=True)  # long queries: hundreds of lists reach the popular rows
    check(oracle, ptr, idx, val, 20, s.search(qp, qi, qv, 20), rows_of(qp, qi, qv, range(12)), [-1] * 12)


def test_tiny_values_take_the_non_atomic_form(oracle):
    """products below 2^-100: partial sums may be subnormal; the library then accumulates by load / add / store (its own
    choice, no hook) and the bits still equal the oracle's, subnormal results included"""
    rng = np.random.default_rng(77)
    ptr, idx, val = random_csr(rng, 500, 40, 1, 10, neg=True)
    val = (val * np.float32(2.0 ** -70)).astype(np.float32)
    s = capi.Sparse(ptr, idx, val)
    got = s.all_pairs(30)
    check(oracle, ptr, idx, val, 30, got, rows_of(ptr, idx, val, range(500)), list(range(500)))
    assert (np.abs(got[1][np.isfinite(got[1])]) < 2.0 ** -126).any()  # subnormal scores were produced and ranked


@pytest.mark.parametrize("atomic", [1, 0])
def test_lists_that_share_rows_keep_the_index_order(oracle, atomic, hooks):
    """Dense-ish rows with values over 12 orders of magnitude and both signs: every accumulator is reached by most lists of a
    query and its float32 sum depends on the order of the products.  Posting lists of 10 .. 200 postings per group laid into
    batches of 64 (several lists per batch, shared rows in nearly every batch)."""
    rng = np.random.default_rng(31)
    rows, dims = 700, 150
    ptr = np.zeros(rows + 1, np.int64)
    idx, val = [], []
    for r in range(rows):
        have = np.nonzero(rng.random(dims) < (0.9 if r < 40 else 0.15))[0]  # 40 rows nearly every list holds
        ptr[r + 1] = ptr[r] + have.size
        idx.append(have.astype(np.uint32))
        val.append((np.exp(rng.uniform(-14, 14, have.size)) * rng.choice([-1.0, 1.0], have.size)).astype(np.float32))
    idx, val = np.concatenate(idx), np.concatenate(val)
    hooks.gorse_hip_test_set_sparse_atomic(atomic)
    # groups of 256 / one group / long queries as one item per group / ... and row by row against the dense query; then every
    # group through the super-visit loop (head 0: the 40 popular rows overflow the hashed table and fall back to the direct
    # accumulators, the others are hashed) and only the first group visited directly
    for tile, split, heavy, head in ((256, 0, 0, 1000), (2048, 0, 0, -1), (256, 64, 0, -1), (256, 20, 21, -1), (2048, 1, 1, -1),
                                     (256, 0, 0, 0), (256, 0, 0, 1), (512, 0, 0, 0)):
        hooks.gorse_hip_test_set_sparse_head(head)
        hooks.gorse_hip_test_set_sparse_tile(tile)
        hooks.gorse_hip_test_set_sparse_split(split)
        hooks.gorse_hip_test_set_sparse_heavy(heavy)
        s = capi.Sparse(ptr, idx, val)
        got = s.all_pairs(50)
        check(oracle, ptr, idx, val, 50, got, rows_of(ptr, idx, val, range(rows)), list(range(rows)))
        s.close()


def test_random_configurations(oracle, hooks):
    """thirty random small problems, each with random k, mask, exclusions and random settings of the library's switches
    (group height, split threshold, workgroups per launch, accumulation form): every answer equals the oracle's"""
    rng = np.random.default_rng(2027)
    for case in range(30):
        rows, dims = int(rng.integers(1, 400 if case % 3 else 3000)), int(rng.integers(1, 120))
        hi = int(rng.integers(0, min(dims, 30) + 1))
        ptr, idx, val = random_csr(rng, rows, dims, 0, hi, neg=bool(rng.integers(0, 2)), zipf=bool(rng.integers(0, 2)))
        hooks.gorse_hip_test_set_sparse_head(int(rng.choice([-1, 0, 1, 2, 1000])))
        hooks.gorse_hip_test_set_sparse_tile(int(rng.choice([0, 256, 512, 2048])))
        hooks.gorse_hip_test_set_sparse_split(int(rng.choice([0, 1, 3, 8, 2048])))
        hooks.gorse_hip_test_set_sparse_heavy(int(rng.choice([0, 2, 5, 12, 16384])))
        hooks.gorse_hip_test_set_sparse_slots(int(rng.choice([0, 1, 2, 5, 64])))
        hooks.gorse_hip_test_set_sparse_atomic(int(rng.choice([-1, 0, 1])))
        s = capi.Sparse(ptr, idx, val)
        k = int(rng.choice([1, 2, 7, 64, 65, 300]))
        mask = None
        if rng.random() < 0.5:
            mask = (rng.random(rows) < rng.random()).astype(np.uint8)  # anything from nearly all hidden to all visible
            s.set_mask(mask)
        if rng.random() < 0.5:
            q0 = int(rng.integers(0, rows))
            q1 = int(rng.integers(q0, rows + 1))
            self_out = bool(rng.integers(0, 2))
            got = s.all_pairs(k, q0, q1, exclude_self=self_out)
            if q1 > q0:
                check(oracle, ptr, idx, val, k, got, rows_of(ptr, idx, val, range(q0, q1)),
                      list(range(q0, q1)) if self_out else [-1] * (q1 - q0), mask)
        else:
            nq = int(rng.integers(1, 40))
            qp, qi, qv = random_csr(rng, nq, dims + 5, 0, min(dims + 5, 40), neg=True)
            excl = rng.integers(-1, rows, nq).astype(np.int64) if rng.random() < 0.5 else None
            got = s.search(qp, qi, qv, k, exclude=excl)
            check(oracle, ptr, idx, val, k, got, rows_of(qp, qi, qv, range(nq)), list(excl) if excl is not None else [-1] * nq, mask)
        s.close()


def test_edge_inputs(oracle, hooks):
    """one stored row without entries; only empty rows; indptr that does not start at 0; an index space with huge gaps; k = 1024;
    an empty query between two others; infinities and a NaN among the values (ordered like the oracle's total order)"""
    s = capi.Sparse(np.array([0, 0], np.int64), np.zeros(0, np.uint32), np.zeros(0, np.float32))
    assert s.all_pairs(3)[2].tolist() == [0]
    assert s.search(np.array([0, 1], np.int64), np.array([5], np.uint32), np.ones(1, np.float32), 2)[2].tolist() == [0]
    s = capi.Sparse(np.zeros(6, np.int64), np.zeros(0, np.uint32), np.zeros(0, np.float32))
    i, sc, c = s.all_pairs(4)
    assert (c == 0).all() and (i == -1).all() and np.isneginf(sc).all()
    rng = np.random.default_rng(1)
    ptr, idx, val = random_csr(rng, 300, 50, 1, 6)
    idx = (idx.astype(np.uint64) * 80000 + 7).astype(np.uint32)  # largest index ~ 4e6
    s = capi.Sparse(ptr + 5, np.concatenate([np.zeros(5, np.uint32), idx]), np.concatenate([np.zeros(5, np.float32), val]))
    check(oracle, ptr, idx, val, 1024, s.all_pairs(1024), rows_of(ptr, idx, val, range(300)), list(range(300)))
    qp = np.array([3, 5, 5, 9], np.int64)
    qi = np.concatenate([np.zeros(3, np.uint32), np.sort(rng.choice(idx, 2, replace=False)), np.sort(rng.choice(np.unique(idx), 4, replace=False))])
    qv = np.ones(9, np.float32)
    check(oracle, ptr, idx, val, 7, s.search(qp, qi.astype(np.uint32), qv, 7), [(qi[3:5], qv[3:5]), (qi[5:5], qv[5:5]), (qi[5:9], qv[5:9])], [-1] * 3)
    ptr3, idx3 = np.array([0, 2, 3, 4], np.int64), np.array([1, 2, 1, 2], np.uint32)
    val3 = np.array([np.inf, 1, np.nan, -np.inf], np.float32)
    got = capi.Sparse(ptr3, idx3, val3).search(np.array([0, 2], np.int64), np.array([1, 2], np.uint32), np.ones(2, np.float32), 5)
    check(oracle, ptr3, idx3, val3, 5, got, [(np.array([1, 2], np.uint32), np.ones(2, np.float32))], [-1])
    assert got[0][0, :3].tolist() == [1, 0, 2]  # NaN (positive sign bit pattern) above +inf above -inf
    # the same through the row-by-row path of the heavy queries (a query value of 0 x inf, a query index nobody stores)
    hooks.gorse_hip_test_set_sparse_split(1)
    hooks.gorse_hip_test_set_sparse_heavy(1)
    qi4, qv4 = np.array([1, 2, 9], np.uint32), np.array([0, 1, 3], np.float32)
    got = capi.Sparse(ptr3, idx3, val3).search(np.array([0, 3], np.int64), qi4, qv4, 5)
    check(oracle, ptr3, idx3, val3, 5, got, [(qi4, qv4)], [-1])
    s = capi.Sparse(ptr + 5, np.concatenate([np.zeros(5, np.uint32), idx]), np.concatenate([np.zeros(5, np.float32), val]))
    check(oracle, ptr, idx, val, 1024, s.all_pairs(1024), rows_of(ptr, idx, val, range(300)), list(range(300)))


def test_argument_errors():
    ptr = np.array([0, 2, 3], np.int64)
    idx = np.array([1, 5, 2], np.uint32)
    val = np.ones(3, np.float32)
    s = capi.Sparse(ptr, idx, val)
    q = (np.array([0, 2], np.int64), np.array([5, 1], np.uint32), np.ones(2, np.float32))
    with pytest.raises(capi.GorseHipError) as e:  # query indices not ascending
        s.search(*q, 3)
    assert e.value.code == capi.ERR_INVALID
    q = (np.array([0, 2], np.int64), np.array([1, 5], np.uint32), np.ones(2, np.float32))
    with pytest.raises(capi.GorseHipError) as e:
        s.search(*q, 1025)
    assert e.value.code == capi.ERR_INVALID
    with pytest.raises(capi.GorseHipError) as e:
        s.search(*q, 3, exclude=[2])
    assert e.value.code == capi.ERR_RANGE
    with pytest.raises(capi.GorseHipError) as e:
        s.all_pairs(3, q_begin=1, q_end=5)
    assert e.value.code == capi.ERR_RANGE
    i, sc, cnt = s.search(*q, 3, exclude=[0])
    assert cnt[0] == 0  # row 1 = {2} shares nothing, row 0 is excluded


# ---- the same kernel behind the vectors.Database twin and the sparse similarity kinds of logics -----------------------
@pytest.mark.parametrize("case", S.SPARSE_CASES + S.PENDING_DENSE_CASES + [S.collaborative_recommend], ids=lambda f: f.__name__)
def test_reference_suite_sparse(case):
    """storage/vectors/database_test.go TestSparse and logics/{item_to_item,user_to_user}_test.go TestTags / TestUsers /
    TestItems / TestAuto (tests/vectors_suite.py) on `vectors.Open("hip://")`"""
    case(V.Open("hip://"))
