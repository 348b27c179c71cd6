"""Inputs and the oracle comparison of the sparse top-k GPU parity tests (tests/test_gpu_vectors_sparse.py)."""
import numpy as np


def random_csr(rng, rows, dims, lo, hi, neg=False, zipf=False):
    """rows sparse vectors over `dims` indices with lo..hi entries each, indices strictly ascending"""
    ptr = [0]
    idx, val = [], []
    pop = None
    if zipf:
        pop = 1.0 / np.arange(1, dims + 1)
        pop = pop / pop.sum()
    for _ in range(rows):
        n = min(int(rng.integers(lo, hi + 1)), dims)
        ids = np.sort(rng.choice(dims, size=n, replace=False, p=pop)).astype(np.uint32)
        v = rng.random(n).astype(np.float32) + np.float32(0.05)
        if neg:
            v *= rng.choice(np.array([-1, 1], dtype=np.float32), size=n)
        idx.append(ids)
        val.append(v)
        ptr.append(ptr[-1] + n)
    return (np.array(ptr, dtype=np.int64), np.concatenate(idx) if idx else np.zeros(0, np.uint32),
            np.concatenate(val) if val else np.zeros(0, np.float32))


def rows_of(ptr, idx, val, rows):
    return [(idx[ptr[r]:ptr[r + 1]], val[ptr[r]:ptr[r + 1]]) for r in rows]


def check_against_oracle(o, ptr, idx, val, k, got, queries, excludes, mask=None):
    """got = (idx nq x k, score nq x k, cnt nq): rows, score BITS, counts and padding equal the oracle's"""
    out_idx, out_sc, out_cnt = got[0], got[1], got[2]
    for t, (qi, qv) in enumerate(queries):
        ei, es = o.sparse_search(ptr, idx, val, qi, qv, k, exclude=excludes[t], admissible=mask)
        assert out_cnt[t] == ei.size, (t, out_cnt[t], ei.size)
        assert np.array_equal(out_idx[t, :ei.size], ei), t
        assert np.array_equal(out_sc[t, :ei.size].view(np.uint32), es.view(np.uint32)), t
        assert (out_idx[t, ei.size:] == -1).all() and np.isneginf(out_sc[t, ei.size:]).all(), t


# rows with equal scores, products that cancel to +-0, an empty row, a row of explicit zeros, negative scores
TIE_ROWS = [([1, 4], [1.0, 1.0]), ([1, 4], [1.0, 1.0]), ([4], [2.0]), ([1, 9], [3.0, 1.0]), ([1, 4], [1.0, -1.0]),
            ([2], [5.0]), ([], []), ([1, 4], [-1.0, 1.0]), ([4, 9], [0.0, 0.0]), ([1], [-4.0]), ([1, 4], [1.0, 1.0]),
            ([4], [-4.0])]
# query 0 = {1: 1, 4: 1} with row 0 excluded and row 2 masked: scores 3 (row 3), 2 (rows 1 and 10: a tie, ascending row),
# -4 (rows 9 and 11), and six zero-score rows (4, 7, 8 cancel or multiply zeros; 5, 6 share nothing) -- the reference
# ranks all ten admissible rows, cuts to k and drops the zeros, so a negative row needs k > 3 + 5
TIE_EXPECT = {3: [3, 1, 10], 8: [3, 1, 10], 9: [3, 1, 10, 9], 10: [3, 1, 10, 9, 11], 20: [3, 1, 10, 9, 11]}


def tie_case():
    ptr = np.array([0] + list(np.cumsum([len(r[0]) for r in TIE_ROWS])), dtype=np.int64)
    idx = np.array([i for r in TIE_ROWS for i in r[0]], dtype=np.uint32)
    val = np.array([v for r in TIE_ROWS for v in r[1]], dtype=np.float32)
    qp = np.array([0, 2, 2, 3], dtype=np.int64)
    qi = np.array([1, 4, 77], dtype=np.uint32)  # query 1 is empty, query 2 only has an index nobody stored
    qv = np.array([1.0, 1.0, 1.0], dtype=np.float32)
    mask = np.ones(len(TIE_ROWS), dtype=np.uint8)
    mask[2] = 0
    excl = np.array([0, -1, -1], dtype=np.int64)
    return ptr, idx, val, (qp, qi, qv), mask, excl
