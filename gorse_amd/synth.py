"""Shape-faithful synthetic stand-ins for the reference's datasets.

MovieLens is fetched from cdn.gorse.io by the reference (model/built_in.go:46-83) and is not
available offline, so tests and bench.py use seeded synthetic data of the same shape:
item popularity Zipf(s) over a random permutation, user activity log-normal, distinct (u, i)
pairs, one held-out positive per user plus `n_neg` fixed negatives (the NCF test.txt layout,
dataset/dataset.go:466-490).  All randomness is numpy's PCG64 seeded explicitly, so the
same arrays are produced here and on the GPU box.
"""
import os
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

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


def _shard_arrays(args):
    rank, world, U_total, I, N_total, seed = args
    d = s_big_shard(rank, world, U_total, I, N_total, seed)
    return d.uptr, d.uidx


def _cache_dir():
    d = os.environ.get("GORSE_SYNTH_CACHE", os.path.join("/tmp", "gorse_synth_cache_%d" % os.getuid()))
    os.makedirs(d, exist_ok=True)
    return d


def s_big_full(world=8, U_total=1_000_000, I=200_000, N_total=100_000_000, seed=43, cache=True):
    """S-big whole (BASELINE config C3 on ONE GPU): the `world` user shards of s_big_shard laid end to end -- users
    [r * U_total / world, (r + 1) * U_total / world) are exactly rank r's shard, so the single-GPU run and the 8-rank run train
    on the same feedback.  The shards are generated in parallel processes; the result (400 MB) is kept under
    $GORSE_SYNTH_CACHE (default /tmp) so that bench.py and the tests of one session generate it once.  User-major CSR only."""
    key = os.path.join(_cache_dir(), "sbig_full_w%d_u%d_i%d_n%d_s%d_np%s.npz" % (world, U_total, I, N_total, seed, np.__version__))
    if cache and os.path.exists(key):
        z = np.load(key)
        uptr, uidx = z["uptr"], z["uidx"]
    else:
        jobs = [(r, world, U_total, I, N_total, seed) for r in range(world)]
        with ProcessPoolExecutor(max_workers=min(world, os.cpu_count() or 1)) as ex:
            parts = list(ex.map(_shard_arrays, jobs))
        lens = np.concatenate([np.diff(p[0]) for p in parts])
        uptr = np.zeros(lens.size + 1, np.int64)
        np.cumsum(lens, out=uptr[1:])
        uidx = np.ascontiguousarray(np.concatenate([p[1] for p in parts]))
        if cache:
            tmp = key + ".tmp%d.npz" % os.getpid()
            np.savez(tmp, uptr=uptr, uidx=uidx)
            os.replace(tmp, key)
    z = np.zeros(uptr.size, np.int64)
    e = np.zeros(0, np.int32)
    return CFData(uptr.size - 1, I, uptr, uidx, None, None, z, e, z.copy(), e.copy())


def s_huge(U=10_000_000, I=1_000_000, N=1_250_000_000, seed=46, zipf_s=1.0, threads=None, cache=True):
    """north_star's "10M x 1M x 128 synthetic set" (BASELINE.json; the feedback count is this repo's choice: 100 per user as
    in C3).  Too large for synth_cf's exact-count rejection rounds, so: log-normal user activity (min 1), Zipf(1.0) item
    popularity over a random permutation, every user's items drawn with replacement and de-duplicated -- the count that
    remains (about 0.8 N: 1.0e9 of the 1.25e9 draws) is what the bench line reports.  Rows come out ascending in the item id (the positive pick of
    the BPR sampler is uniform over the row, model.go:459, so the stored order does not matter to training).  Users are
    cut into blocks, one numpy Generator per block (SeedSequence.spawn), blocks on host threads.  User-major CSR only."""
    z = np.zeros(U + 1, np.int64)
    e = np.zeros(0, np.int32)
    key = os.path.join(_cache_dir(), "shuge_u%d_i%d_n%d_s%d_z%g_np%s" % (U, I, N, seed, zipf_s, np.__version__))
    if cache and os.path.exists(key + "_uidx.npy"):
        return CFData(U, I, np.load(key + "_uptr.npy"), np.load(key + "_uidx.npy"), None, None, z, e, z.copy(), e.copy())
    rng = np.random.default_rng(seed)
    act = rng.lognormal(0.0, 1.0, U)
    lens = np.maximum(1, np.floor(act / act.sum() * N)).astype(np.int64)
    w = 1.0 / np.power(np.arange(1, I + 1, dtype=np.float64), zipf_s)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    perm = rng.permutation(I).astype(np.int64)
    block = 200_000
    starts = list(range(0, U, block))
    seeds = np.random.SeedSequence(seed).spawn(len(starts))
    if threads is None:
        try:
            avail_gb = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 2**30
        except (ValueError, OSError):
            avail_gb = 16.0
        threads = int(max(1, min(os.cpu_count() or 1, 32, (avail_gb - 16) // 2)))

    def work(b):
        u0 = starts[b]
        u1 = min(U, u0 + block)
        g = np.random.default_rng(seeds[b])
        ln = lens[u0:u1]
        rows = np.repeat(np.arange(u1 - u0, dtype=np.int64), ln)
        items = perm[np.minimum(np.searchsorted(cdf, g.random(rows.size)), I - 1)]
        key = np.unique(rows * I + items)  # sorted: by user, then item; duplicates of a (user, item) pair once
        rows = key // I
        return np.bincount(rows, minlength=u1 - u0), (key - rows * I).astype(np.int32)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(work, range(len(starts))))
    cnt = np.concatenate([p[0] for p in parts])
    uptr = np.zeros(U + 1, np.int64)
    np.cumsum(cnt, out=uptr[1:])
    uidx = np.empty(int(uptr[-1]), np.int32)
    o = 0
    for _, it in parts:
        uidx[o:o + it.size] = it
        o += it.size
    if cache and U * 1 >= 1_000_000:  # gigabytes: a second bench command of the session (a profiler pass) loads it instead
        np.save(key + "_uptr.npy", uptr)
        tmp = key + "_uidx.tmp%d.npy" % os.getpid()
        np.save(tmp, uidx)
        os.replace(tmp, key + "_uidx.npy")
    return CFData(U, I, uptr, uidx, None, None, z, e, z.copy(), e.copy())


def hold_out(data, n_users, n_neg, seed):
    """Leave-one-out over the FIRST n_users users with >= 2 feedbacks (dataset.go:258-285 does it for every user): the last
    stored feedback becomes the test positive, n_neg items outside the user's feedback the fixed negatives (the NCF test.txt
    layout, dataset.go:466-490).  Returns a CFData whose training CSR lacks the held-out entries (user-major only)."""
    rng = np.random.default_rng(seed)
    lens = np.diff(data.uptr)
    users = np.nonzero(lens >= 2)[0][:n_users]
    keep = np.ones(data.uidx.size, bool)
    last = data.uptr[users + 1] - 1
    keep[last] = False
    newlen = lens.copy()
    newlen[users] -= 1
    uidx = np.ascontiguousarray(data.uidx[keep])
    uptr = np.zeros(data.U + 1, np.int64)
    np.cumsum(newlen, out=uptr[1:])
    has = np.zeros(data.U, bool)
    has[users] = True
    test_ptr = np.zeros(data.U + 1, np.int64)
    np.cumsum(has, out=test_ptr[1:])
    test_idx = np.ascontiguousarray(data.uidx[last].astype(np.int32))
    neg_ptr = np.zeros(data.U + 1, np.int64)
    np.cumsum(np.where(has, n_neg, 0), out=neg_ptr[1:])
    neg_idx = np.empty(users.size * n_neg, np.int32)
    for t, u in enumerate(users):
        posset = data.uidx[data.uptr[u]:data.uptr[u + 1]]
        got = np.empty(0, np.int64)
        while got.size < n_neg:
            c = rng.integers(0, data.I, size=2 * n_neg)
            got = np.unique(np.concatenate([got, c[~np.isin(c, posset)]]))
        neg_idx[t * n_neg:(t + 1) * n_neg] = rng.permutation(got)[:n_neg]
    return CFData(data.U, data.I, uptr, uidx, None, None, test_ptr, test_idx, neg_ptr, neg_idx)


def init_factors_big(U, I, d, mean, std, seed, threads=None):
    """init_factors for matrices of gigabytes: row blocks on host threads, one Generator per block."""
    threads = threads or min(os.cpu_count() or 1, 32)

    def fill(n, sd):
        out = np.empty((n, d), np.float32)
        blk = 1 << 18
        starts = list(range(0, n, blk))
        seeds = np.random.SeedSequence(sd).spawn(len(starts))

        def work(b):
            r0, r1 = starts[b], min(n, starts[b] + blk)
            g = np.random.default_rng(seeds[b])
            g.standard_normal(out=out[r0:r1], dtype=np.float32)
            out[r0:r1] *= np.float32(std)
            out[r0:r1] += np.float32(mean)
        with ThreadPoolExecutor(max_workers=threads) as ex:
            list(ex.map(work, range(len(starts))))
        return out
    return fill(U, seed), fill(I, seed + 1)


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
