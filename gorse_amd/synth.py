"""Shape-faithful synthetic stand-ins for the reference's datasets.

MovieLens is fetched from cdn.gorse.io by the reference (model/built_in.go:46-83) and is not
available offline, so tests and bench.py use seeded synthetic data of the same shape:
item popularity Zipf(s) over a random permutation, user activity log-normal, distinct (u, i)
pairs, one held-out positive per user plus `n_neg` fixed negatives (the NCF test.txt layout,
dataset/dataset.go:466-490).  All randomness is numpy's PCG64 seeded explicitly, so the
same arrays are produced here and on the GPU box.
"""
import numpy as np


class CFData:
    """Flattened dataset.CFSplit pair (train, test) -- dataset/dataset.go:40-59."""

    def __init__(self, U, I, uptr, uidx, iptr, iidx, test_ptr, test_idx, neg_ptr, neg_idx):
        self.U, self.I = int(U), int(I)
        self.uptr, self.uidx, self.iptr, self.iidx = uptr, uidx, iptr, iidx
        self.test_ptr, self.test_idx, self.neg_ptr, self.neg_idx = test_ptr, test_idx, neg_ptr, neg_idx

    @property
    def n_train(self):
        return int(self.uptr[-1])

    def candidates(self):
        """Evaluate's candidate lists (model/cf/evaluator.go:47-53): test positives then negatives,
        for users that have test feedback.  Returns (users, cand_ptr, cand)."""
        tl = np.diff(self.test_ptr)
        nl = np.diff(self.neg_ptr)
        users = np.nonzero(tl > 0)[0].astype(np.int32)
        lens = (tl + nl)[users]
        cptr = np.zeros(users.size + 1, np.int64)
        np.cumsum(lens, out=cptr[1:])
        cand = np.empty(int(cptr[-1]), np.int32)
        for k, u in enumerate(users):  # small (one entry per test user)
            a = cptr[k]
            t = self.test_idx[self.test_ptr[u]:self.test_ptr[u + 1]]
            n = self.neg_idx[self.neg_ptr[u]:self.neg_ptr[u + 1]]
            cand[a:a + t.size] = t
            cand[a + t.size:a + t.size + n.size] = n
        return users, cptr, cand


def _csr_from_pairs(rows, cols, nrows):
    order = np.argsort(rows, kind="stable")
    rows, cols = rows[order], cols[order]
    ptr = np.zeros(nrows + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=nrows), out=ptr[1:])
    return ptr, np.ascontiguousarray(cols.astype(np.int32))


def synth_cf(U, I, N, seed, zipf_s=1.0, sigma=1.0, min_len=2, max_frac=0.5, n_neg=99, with_test=True):
    """U users, I items, exactly-or-nearly N train feedbacks (+ one test positive per user)."""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(I)
    wts = 1.0 / np.power(np.arange(1, I + 1, dtype=np.float64), zipf_s)
    cdf = np.cumsum(wts)
    cdf /= cdf[-1]
    target = N + (U if with_test else 0)
    act = rng.lognormal(0.0, sigma, U)
    hi = max(min_len + 1, int(I * max_frac))
    lens = np.clip(act / act.sum() * target, min_len, hi)
    for _ in range(8):  # rescale after clipping
        lens = np.clip(lens * (target / lens.sum()), min_len, hi)
    lens = np.floor(lens).astype(np.int64)
    deficit = int(target - lens.sum())
    if deficit > 0:
        bump = rng.choice(np.nonzero(lens < hi)[0], size=deficit, replace=True)
        np.add.at(lens, bump, 1)
        lens = np.minimum(lens, hi)
    # draw with replacement, de-duplicate, top up the users still short; draw order = "insertion order"
    heavy = lens > I // 8  # rejection is slow for very active users: sampled without replacement below
    have_u = np.empty(0, np.int64)
    have_i = np.empty(0, np.int64)
    need = np.where(heavy, 0, lens)
    for rnd in range(12):
        cur = np.bincount(have_u, minlength=U) if have_u.size else np.zeros(U, np.int64)
        deficit = need - cur
        if not (deficit > 0).any():
            break
        over = np.where(deficit > 0, (deficit * (1.3 + 0.2 * rnd)).astype(np.int64) + 4, 0)
        urep = np.repeat(np.arange(U, dtype=np.int64), over)
        items = perm[np.searchsorted(cdf, rng.random(urep.size), side="right").clip(0, I - 1)]
        all_u = np.concatenate([have_u, urep])
        all_i = np.concatenate([have_i, items])
        key = all_u * I + all_i
        _, first = np.unique(key, return_index=True)
        first.sort()  # earlier draws win, order of first appearance is kept
        all_u, all_i = all_u[first], all_i[first]
        order = np.argsort(all_u, kind="stable")
        all_u, all_i = all_u[order], all_i[order]
        ptr = np.zeros(U + 1, np.int64)
        np.cumsum(np.bincount(all_u, minlength=U), out=ptr[1:])
        pos = np.arange(all_u.size) - ptr[all_u]
        keep = pos < need[all_u]
        have_u, have_i = all_u[keep], all_i[keep]
    urep, items = have_u, have_i
    if heavy.any():
        hu, hi_items = [], []
        p = wts[np.argsort(perm)]
        p = p / p.sum()
        for u in np.nonzero(heavy)[0]:
            it = rng.choice(I, size=int(lens[u]), replace=False, p=p)
            hu.append(np.full(it.size, u, np.int64))
            hi_items.append(it)
        urep = np.concatenate([urep] + hu)
        items = np.concatenate([items] + hi_items)
    uptr_all, uidx_all = _csr_from_pairs(urep, items, U)
    if not with_test:
        iptr, iidx = _csr_from_pairs(uidx_all.astype(np.int64), np.repeat(np.arange(U), np.diff(uptr_all)), I)
        z = np.zeros(U + 1, np.int64)
        e = np.zeros(0, np.int32)
        return CFData(U, I, uptr_all, uidx_all, iptr, iidx, z, e, z.copy(), e.copy())
    # leave-one-out: the LAST stored feedback of every user with >= 2 feedbacks goes to the test split
    ulen = np.diff(uptr_all)
    has_test = ulen >= 2
    last = uptr_all[1:] - 1
    mask = np.ones(uidx_all.size, bool)
    mask[last[has_test]] = False
    test_idx = uidx_all[last[has_test]].astype(np.int32)
    test_ptr = np.zeros(U + 1, np.int64)
    np.cumsum(has_test.astype(np.int64), out=test_ptr[1:])
    rows = np.repeat(np.arange(U, dtype=np.int64), ulen)[mask]
    cols = uidx_all[mask]
    uptr, uidx = _csr_from_pairs(rows, cols, U)
    iptr, iidx = _csr_from_pairs(cols.astype(np.int64), rows, I)
    # negatives: n_neg distinct items per test user outside the user's positives (train + test)
    neg_ptr = np.zeros(U + 1, np.int64)
    np.cumsum(np.where(has_test, n_neg, 0), out=neg_ptr[1:])
    neg_idx = np.empty(int(neg_ptr[-1]), np.int32)
    for u in np.nonzero(has_test)[0]:
        posset = uidx_all[uptr_all[u]:uptr_all[u + 1]]
        avail = I - posset.size
        k = min(n_neg, avail)
        got = np.empty(0, np.int64)
        while got.size < k:
            c = rng.integers(0, I, size=2 * (k - got.size) + 8)
            c = c[~np.isin(c, posset)]
            got = np.unique(np.concatenate([got, c]))
            if got.size > k:
                got = rng.permutation(got)[:k]
        got = rng.permutation(got)[:k]
        a = neg_ptr[u]
        neg_idx[a:a + k] = got
        if k < n_neg:  # cannot happen for the shapes used; keep the CSR consistent anyway
            neg_idx[a + k:a + n_neg] = got[:1].repeat(n_neg - k) if k > 0 else 0
    return CFData(U, I, uptr, uidx, iptr, iidx, test_ptr, test_idx, neg_ptr, neg_idx)


# the named synthetic configurations of SURVEY.md section 8d / BASELINE.md section 3
def s_ml100k():
    return synth_cf(943, 1682, 99057, seed=42, min_len=19, n_neg=99)


def s_ml1m():
    return synth_cf(6040, 3706, 994169, seed=42, min_len=19, n_neg=99)


def s_big_shard(rank=0, world=8, U_total=1_000_000, I=200_000, N_total=100_000_000, seed=43):
    """One rank's user shard of S-big (C3): U_total/world users, N_total/world feedbacks, all I items."""
    return synth_cf(U_total // world, I, N_total // world, seed=seed + 1000 * rank, min_len=1, n_neg=99, with_test=False)


def init_factors(U, I, d, mean, std, seed):
    """BaseMatrixFactorization init (model/cf/model.go:532-540; util/random.go:45-60): users first,
    then items, row-major, float32(N(0,1)) * std + mean.  Go's math/rand stream is not reproducible
    here, so only the distribution and the draw order are kept."""
    rng = np.random.default_rng(seed)
    P = (rng.standard_normal((U, d)).astype(np.float32) * np.float32(std) + np.float32(mean)).astype(np.float32)
    Q = (rng.standard_normal((I, d)).astype(np.float32) * np.float32(std) + np.float32(mean)).astype(np.float32)
    return P, Q


def s_emb(N=1_000_000, d=128, seed=44):
    """S-emb (BASELINE config C4): N x d entries N(0,1), L2-normalised in fp32, then bf16 by
    truncation (common/bfloats/bfloats.go:24-30).  Returns (uint16 N x d, the same values expanded to fp32)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, d), dtype=np.float32)
    X /= np.sqrt(np.einsum("ij,ij->i", X, X, dtype=np.float32))[:, None]
    Xb = (X.view(np.uint32) >> 16).astype(np.uint16)
    return Xb, (Xb.astype(np.uint32) << 16).view(np.float32)


def s_als(U=500_000, I=100_000, nnz=50_000_000, seed=45, zipf_s=1.0):
    """S-als (BASELINE config C5; nnz is this repo's assumption, BASELINE gives none): log-normal user
    activity (min 1), Zipf(1.0) item popularity over a random permutation, drawn with replacement -- a pair may
    repeat inside a row, which the ALS kernels treat as two feedback entries (timing input, not a parity fixture).
    Returns (uptr, uidx, iptr, iidx): user-major and item-major CSR of the same entries."""
    rng = np.random.default_rng(seed)
    act = rng.lognormal(0.0, 1.0, U)
    lens = np.maximum(1, np.floor(act / act.sum() * nnz)).astype(np.int64)
    uptr = np.zeros(U + 1, np.int64)
    np.cumsum(lens, out=uptr[1:])
    n = int(uptr[-1])
    w = 1.0 / np.power(np.arange(1, I + 1, dtype=np.float64), zipf_s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    perm = rng.permutation(I).astype(np.int32)
    uidx = perm[np.minimum(np.searchsorted(cdf, rng.random(n)), I - 1)].astype(np.int32)
    rows = np.repeat(np.arange(U, dtype=np.int32), lens)
    order = np.argsort(uidx, kind="stable")
    iidx = rows[order].astype(np.int32)
    iptr = np.zeros(I + 1, np.int64)
    np.cumsum(np.bincount(uidx, minlength=I), out=iptr[1:])
    return uptr, uidx, iptr, iidx


def idf_vectors(ptr, idx, n_other):
    """The sparse vectors of the "users" item-to-item / "items" user-to-user recommenders for the rows of one CSR
    side (logics/item_to_item.go:209-220, user_to_user.go:201-212): row r = its feedback ids ascending
    (slices.Sort), value sqrt(idf[id]) with idf[id] = log(1 + n_rows / freq(id)) in float32
    (dataset/dataset.go:160-180: GetUserIDF divides the ITEM count by the user's frequency and vice versa; here the
    rows are the counted side), ids with idf <= 0 dropped (vector_writer.go:200-208).
    ptr/idx: row -> ids CSR (e.g. CFData.iptr/iidx = item -> users); n_other = number of distinct ids.
    Returns (indptr int64, indices uint32, values float32)."""
    ptr = np.asarray(ptr, np.int64)
    idx = np.asarray(idx, np.int64)
    n_rows = ptr.size - 1
    freq = np.bincount(idx, minlength=n_other).astype(np.float32)
    with np.errstate(divide="ignore"):
        idf = np.log((np.float32(1) + np.float32(n_rows) / freq).astype(np.float64)).astype(np.float32)
    val_of = np.sqrt(idf.astype(np.float64)).astype(np.float32)
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), np.diff(ptr))
    order = np.lexsort((idx, rows))
    rows, ids = rows[order], idx[order]
    keep = idf[ids] > 0
    keep[1:] &= ~((rows[1:] == rows[:-1]) & (ids[1:] == ids[:-1]))  # a set: duplicates once
    rows, ids = rows[keep], ids[keep]
    out_ptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=n_rows), out=out_ptr[1:])
    return out_ptr, np.ascontiguousarray(ids.astype(np.uint32)), np.ascontiguousarray(val_of[ids])
