"""ctypes/numpy front-end of the CPU ORACLE (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (gorse_amd) never does.

`Oracle` wraps oracle/liboracle.so (our C restatement, gorse_oracle.c).
`Ref` wraps oracle/_ref/libgorse_ref.so (the reference's own SIMD C kernels,
compiled from /root/reference by oracle/Makefile) and is None-safe: on a host
without AVX512 or without the prebuilt file, `load_ref()` returns None.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ISA_GO, ISA_AVX, ISA_AVX512 = 0, 1, 2
METRIC_NEG_DOT, METRIC_EUCLIDEAN, METRIC_COSINE, METRIC_EUCLIDEAN_BF16 = 0, 1, 2, 3
M_NDCG, M_PRECISION, M_RECALL, M_HR, M_MAP, M_MRR = range(6)

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_i64p)


def build(force=False):
    """Compile liboracle.so (always possible) and _ref (only where /root/reference exists)."""
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "gorse_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE], stdout=subprocess.DEVNULL)
    ref = os.path.join(HERE, "_ref", "libgorse_ref.so")
    if os.path.isdir("/root/reference/common/floats/src") and (force or not os.path.exists(ref)):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


class SparseIndex:
    """orc_sparse_index_*: posting lists of the stored rows; search() = orc_sparse_search_inverted (same results as
    Oracle.sparse_search, found by walking posting lists).  scratch() makes the per-thread work arrays."""

    def __init__(self, oracle, indptr, indices, values):
        self.o = oracle
        self.indptr, pp = _i64(indptr)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.values, pv = _f32(values)
        self.N = self.indptr.size - 1
        self.h = oracle.L.orc_sparse_index_build(self.N, pp, self.indices.ctypes.data_as(_u32p), pv)

    def close(self):
        if getattr(self, "h", None):
            self.o.L.orc_sparse_index_free(self.h)
            self.h = None

    __del__ = close

    def scratch(self):
        return np.zeros(self.N, np.float32), np.zeros(self.N, np.uint8), np.zeros(self.N, np.int32)

    def search(self, q_idx, q_val, k, exclude=-1, admissible=None, scratch=None):
        """(row indices, scores, postings walked)"""
        acc, seen, touched = scratch if scratch is not None else self.scratch()
        q_idx = np.ascontiguousarray(q_idx, dtype=np.uint32)
        q_val, pqv = _f32(q_val)
        pm, n_adm = None, self.N
        if admissible is not None:
            admissible = np.ascontiguousarray(admissible, dtype=np.uint8)
            pm = admissible.ctypes.data_as(C.POINTER(C.c_uint8))
            n_adm = int(np.count_nonzero(admissible))
        oi = np.zeros(k + 1, dtype=np.int32)
        ow = np.zeros(k + 1, dtype=np.float32)
        walked = C.c_int64(0)
        cnt = self.o.L.orc_sparse_search_inverted(self.h, self.N, q_idx.ctypes.data_as(_u32p), pqv, q_idx.size, int(exclude), pm,
                                                  n_adm, k, acc.ctypes.data_as(_f32p), seen.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                  touched.ctypes.data_as(_i32p), oi.ctypes.data_as(_i32p),
                                                  ow.ctypes.data_as(_f32p), C.byref(walked))
        return oi[:cnt].copy(), ow[:cnt].copy(), walked.value


class Oracle:
    def __init__(self):
        build()
        L = C.CDLL(os.path.join(HERE, "liboracle.so"))
        self.L = L
        L.orc_dot.restype = C.c_float
        L.orc_dot.argtypes = [_f32p, _f32p, C.c_int64]
        L.orc_euclidean.restype = C.c_float
        L.orc_euclidean.argtypes = [_f32p, _f32p, C.c_int64]
        L.orc_bf16_euclidean.restype = C.c_float
        L.orc_bf16_euclidean.argtypes = [_u16p, _u16p, C.c_int64]
        L.orc_exp.restype = C.c_float
        L.orc_exp.argtypes = [C.c_float]
        L.orc_exp_restated.restype = C.c_float
        L.orc_exp_restated.argtypes = [C.c_float]
        L.orc_distance.restype = C.c_float
        L.orc_distance.argtypes = [C.c_int, _f32p, _f32p, C.c_int64]
        L.orc_metric.restype = C.c_float
        L.orc_metric.argtypes = [C.c_int, _i32p, C.c_int64, _i32p, C.c_int64]
        L.orc_bpr_apply_triplets.restype = C.c_double
        L.orc_bpr_apply_triplets.argtypes = [_f32p, _f32p, C.c_int64, _i32p, _i32p, _i32p, C.c_int64, C.c_float,
                                             C.c_float]
        L.orc_bpr_epoch_sampled.restype = C.c_double
        L.orc_bpr_epoch_sampled.argtypes = [_f32p, _f32p, C.c_int64, C.c_int64, C.c_int64, _i64p, _i32p, _i32p,
                                            C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_float, C.c_float]
        L.orc_sample_user_negatives.restype = None
        L.orc_sample_user_negatives.argtypes = [C.c_int64, C.c_int64, _i64p, _i32p, _i64p, _i32p, C.c_int32, C.c_uint64, _i32p, _i32p]
        L.orc_bpr_sample.restype = None
        L.orc_bpr_sample.argtypes = [C.c_int64, C.c_int64, _i64p, _i32p, _i32p, C.c_uint64, C.c_uint64, C.c_int64,
                                     C.c_int64, _i32p, _i32p, _i32p]
        L.orc_als_half_range.restype = None
        L.orc_als_half_range.argtypes = [_f32p, _f32p, C.c_int64, C.c_int64, C.c_int64, _i64p, _i32p, _i64p, C.c_float,
                                         C.c_float, C.c_int64, C.c_int64]
        L.orc_als_epoch.restype = None
        L.orc_als_epoch.argtypes = [_f32p, _f32p, C.c_int64, C.c_int64, C.c_int64, _i64p, _i32p, _i64p, _i32p,
                                    C.c_float, C.c_float]
        L.orc_mf_score.restype = None
        L.orc_mf_score.argtypes = [_f32p, _f32p, C.c_int64, _i32p, _i32p, C.c_int64, _f32p]
        L.orc_mf_rank.restype = None
        L.orc_mf_rank.argtypes = [_f32p, _f32p, C.c_int64, C.c_int64, _i32p, _i64p, _i32p, C.c_int, _i32p, _i32p]
        L.orc_evaluate.restype = None
        L.orc_evaluate.argtypes = [_f32p, _f32p, C.c_int64, C.c_int64, _i64p, _i32p, _i64p, _i32p, C.c_int, _i32p,
                                   C.c_int, _f32p]
        L.orc_topk_filter.restype = C.c_int
        L.orc_topk_filter.argtypes = [C.c_int, C.c_int64, _i32p, _f32p, _i32p, _f32p]
        L.orc_pq_sort.restype = C.c_int
        L.orc_pq_sort.argtypes = [C.c_int, C.c_int64, _i32p, _f32p, _i32p, _f32p]
        L.orc_bruteforce_select.restype = C.c_int
        L.orc_bruteforce_select.argtypes = [C.c_int64, _f32p, C.c_int64, C.c_int, C.c_int, _i32p, _f32p]
        L.orc_bruteforce_search_index.restype = C.c_int
        L.orc_bruteforce_search_index.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int,
                                                  _i32p, _f32p]
        L.orc_bruteforce_search_vector.restype = C.c_int
        L.orc_bruteforce_search_vector.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int, _f32p, C.c_int, C.c_int,
                                                   _i32p, _f32p]
        L.orc_philox4x32_10.restype = None
        L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        _u8p = C.POINTER(C.c_uint8)
        L.orc_sparse_dot.restype = C.c_int64
        L.orc_sparse_dot.argtypes = [_u32p, _f32p, C.c_int64, _u32p, _f32p, C.c_int64, _f32p]
        L.orc_sparse_search.restype = C.c_int
        L.orc_sparse_search.argtypes = [C.c_int64, _i64p, _u32p, _f32p, _u32p, _f32p, C.c_int64, C.c_int64, _u8p,
                                        C.c_int, _i32p, _f32p]
        L.orc_sparse_index_build.restype = C.c_void_p
        L.orc_sparse_index_build.argtypes = [C.c_int64, _i64p, _u32p, _f32p]
        L.orc_sparse_index_free.restype = None
        L.orc_sparse_index_free.argtypes = [C.c_void_p]
        L.orc_sparse_search_inverted.restype = C.c_int
        L.orc_sparse_search_inverted.argtypes = [C.c_void_p, C.c_int64, _u32p, _f32p, C.c_int64, C.c_int64, _u8p, C.c_int64,
                                                 C.c_int, _f32p, _u8p, _i32p, _i32p, _f32p, _i64p]
        L.orc_idf.restype = None
        L.orc_idf.argtypes = [_i32p, C.c_int64, C.c_int64, _f32p]
        L.orc_mm.restype = None
        L.orc_mm.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, _f32p, C.c_int64, _f32p, C.c_int64,
                             _f32p, C.c_int64]

    # ---- configuration -------------------------------------------------
    def set_isa(self, isa):
        self.L.orc_set_isa(int(isa))

    def set_exp(self, mode):
        self.L.orc_set_exp(int(mode))

    # ---- floats --------------------------------------------------------
    def dot(self, a, b):
        a, pa = _f32(a)
        b, pb = _f32(b)
        assert a.size == b.size
        return float(np.float32(self.L.orc_dot(pa, pb, a.size)))

    def euclidean(self, a, b):
        a, pa = _f32(a)
        b, pb = _f32(b)
        return float(np.float32(self.L.orc_euclidean(pa, pb, a.size)))

    def _vec(self, name, *args):
        getattr(self.L, name)(*args)

    def mul_const_add(self, a, c, dst):
        a, pa = _f32(a)
        dst = np.array(dst, dtype=np.float32)
        self.L.orc_mul_const_add.argtypes = [_f32p, C.c_float, _f32p, C.c_int64]
        self.L.orc_mul_const_add(pa, c, dst.ctypes.data_as(_f32p), a.size)
        return dst

    def mul_const_add_to(self, a, b, c):
        a, pa = _f32(a)
        c, pc = _f32(c)
        dst = np.zeros_like(a)
        self.L.orc_mul_const_add_to.argtypes = [_f32p, C.c_float, _f32p, _f32p, C.c_int64]
        self.L.orc_mul_const_add_to(pa, b, pc, dst.ctypes.data_as(_f32p), a.size)
        return dst

    def mul_const_to(self, a, b):
        a, pa = _f32(a)
        dst = np.zeros_like(a)
        self.L.orc_mul_const_to.argtypes = [_f32p, C.c_float, _f32p, C.c_int64]
        self.L.orc_mul_const_to(pa, b, dst.ctypes.data_as(_f32p), a.size)
        return dst

    def sub_to(self, a, b):
        a, pa = _f32(a)
        b, pb = _f32(b)
        dst = np.zeros_like(a)
        self.L.orc_sub_to.argtypes = [_f32p, _f32p, _f32p, C.c_int64]
        self.L.orc_sub_to(pa, pb, dst.ctypes.data_as(_f32p), a.size)
        return dst

    def mm(self, transA, transB, m, n, k, a, lda, b, ldb, c, ldc):
        a, pa = _f32(a)
        b, pb = _f32(b)
        c = np.array(c, dtype=np.float32)
        self.L.orc_mm(int(transA), int(transB), m, n, k, pa, lda, pb, ldb, c.ctypes.data_as(_f32p), ldc)
        return c

    # ---- bf16 ----------------------------------------------------------
    def bf16_from_f32(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.zeros(a.shape, dtype=np.uint16)
        self.L.orc_bf16_from_f32.argtypes = [_f32p, _u16p, C.c_int64]
        self.L.orc_bf16_from_f32(a.ctypes.data_as(_f32p), out.ctypes.data_as(_u16p), a.size)
        return out

    def bf16_to_f32(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint16)
        out = np.zeros(a.shape, dtype=np.float32)
        self.L.orc_bf16_to_f32.argtypes = [_u16p, _f32p, C.c_int64]
        self.L.orc_bf16_to_f32(a.ctypes.data_as(_u16p), out.ctypes.data_as(_f32p), a.size)
        return out

    def bf16_euclidean(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint16)
        b = np.ascontiguousarray(b, dtype=np.uint16)
        return float(np.float32(self.L.orc_bf16_euclidean(a.ctypes.data_as(_u16p), b.ctypes.data_as(_u16p), a.size)))

    # ---- heaps ---------------------------------------------------------
    def topk_filter(self, k, items, weights):
        items, pi = _i32(items)
        weights, pw = _f32(weights)
        oi = np.zeros(max(k, 1), dtype=np.int32)
        ow = np.zeros(max(k, 1), dtype=np.float32)
        n = self.L.orc_topk_filter(k, items.size, pi, pw, oi.ctypes.data_as(_i32p), ow.ctypes.data_as(_f32p))
        return oi[:n].copy(), ow[:n].copy()

    def pq_sort(self, desc, items, weights):
        items, pi = _i32(items)
        weights, pw = _f32(weights)
        oi = np.zeros(items.size + 1, dtype=np.int32)
        ow = np.zeros(items.size + 1, dtype=np.float32)
        n = self.L.orc_pq_sort(int(desc), items.size, pi, pw, oi.ctypes.data_as(_i32p), ow.ctypes.data_as(_f32p))
        return oi[:n].copy(), ow[:n].copy()

    def bruteforce_select(self, dist, skip, k, prune0=False):
        dist, pd = _f32(dist)
        oi = np.zeros(k + 1, dtype=np.int32)
        ow = np.zeros(k + 1, dtype=np.float32)
        n = self.L.orc_bruteforce_select(dist.size, pd, skip, k, int(prune0), oi.ctypes.data_as(_i32p),
                                         ow.ctypes.data_as(_f32p))
        return oi[:n].copy(), ow[:n].copy()

    def search_index(self, X, metric, q, k, prune0=False):
        X, px = _f32(X)
        n, d = X.shape
        oi = np.zeros(k + 1, dtype=np.int32)
        ow = np.zeros(k + 1, dtype=np.float32)
        cnt = self.L.orc_bruteforce_search_index(px, n, d, metric, q, k, int(prune0), oi.ctypes.data_as(_i32p),
                                                 ow.ctypes.data_as(_f32p))
        if cnt < 0:
            raise IndexError("index out of range: %d" % q)
        return oi[:cnt].copy(), ow[:cnt].copy()

    def search_vector(self, X, metric, qv, k, prune0=False):
        X, px = _f32(X)
        qv, pq = _f32(qv)
        n, d = X.shape
        oi = np.zeros(k + 1, dtype=np.int32)
        ow = np.zeros(k + 1, dtype=np.float32)
        cnt = self.L.orc_bruteforce_search_vector(px, n, d, metric, pq, k, int(prune0), oi.ctypes.data_as(_i32p),
                                                  ow.ctypes.data_as(_f32p))
        return oi[:cnt].copy(), ow[:cnt].copy()

    # ---- sparse collections (vectors.Database, dimension 0 / Dot) -------------------------
    def sparse_dot(self, ia, va, ib, vb):
        ia = np.ascontiguousarray(ia, dtype=np.uint32)
        ib = np.ascontiguousarray(ib, dtype=np.uint32)
        va, pva = _f32(va)
        vb, pvb = _f32(vb)
        out = np.zeros(1, dtype=np.float32)
        common = self.L.orc_sparse_dot(ia.ctypes.data_as(_u32p), pva, ia.size, ib.ctypes.data_as(_u32p), pvb, ib.size,
                                       out.ctypes.data_as(_f32p))
        return int(common), out[0]

    def sparse_search(self, indptr, indices, values, q_idx, q_val, k, exclude=-1, admissible=None):
        """One query against the CSR rows: (row indices, scores) of the k best hits, best first."""
        indptr, pp = _i64(indptr)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        values, pv = _f32(values)
        q_idx = np.ascontiguousarray(q_idx, dtype=np.uint32)
        q_val, pqv = _f32(q_val)
        pm = None
        if admissible is not None:
            admissible = np.ascontiguousarray(admissible, dtype=np.uint8)
            pm = admissible.ctypes.data_as(C.POINTER(C.c_uint8))
        oi = np.zeros(k + 1, dtype=np.int32)
        ow = np.zeros(k + 1, dtype=np.float32)
        cnt = self.L.orc_sparse_search(indptr.size - 1, pp, indices.ctypes.data_as(_u32p), pv,
                                       q_idx.ctypes.data_as(_u32p), pqv, q_idx.size, int(exclude), pm, k,
                                       oi.ctypes.data_as(_i32p), ow.ctypes.data_as(_f32p))
        return oi[:cnt].copy(), ow[:cnt].copy()

    def sparse_index(self, indptr, indices, values):
        """Inverted index over the CSR rows (the CPU baseline of bench.py's sparse leg); see SparseIndex."""
        return SparseIndex(self, indptr, indices, values)

    def idf(self, freq, total):
        freq, pf = _i32(freq)
        out = np.zeros(freq.size, dtype=np.float32)
        self.L.orc_idf(pf, freq.size, int(total), out.ctypes.data_as(_f32p))
        return out

    def distance(self, metric, a, b):
        a, pa = _f32(a)
        b, pb = _f32(b)
        return float(np.float32(self.L.orc_distance(metric, pa, pb, a.size)))

    # ---- rng / sampling --------------------------------------------------
    def philox(self, ctr, key):
        ctr = np.ascontiguousarray(ctr, dtype=np.uint32)
        key = np.ascontiguousarray(key, dtype=np.uint32)
        out = np.zeros(4, dtype=np.uint32)
        self.L.orc_philox4x32_10(ctr.ctypes.data_as(_u32p), key.ctypes.data_as(_u32p), out.ctypes.data_as(_u32p))
        return out

    def bpr_sample(self, U, I, indptr, indices, n, seed, epoch, sample_base=0, sorted_indices=None):
        indptr, pp = _i64(indptr)
        indices, pi = _i32(indices)
        if sorted_indices is None:
            sorted_indices = sort_rows(indptr, indices)
        sorted_indices, ps = _i32(sorted_indices)
        u = np.zeros(n, dtype=np.int32)
        i = np.zeros(n, dtype=np.int32)
        j = np.zeros(n, dtype=np.int32)
        self.L.orc_bpr_sample(U, I, pp, pi, ps, seed, epoch, sample_base, n, u.ctypes.data_as(_i32p),
                              i.ctypes.data_as(_i32p), j.ctypes.data_as(_i32p))
        return u, i, j

    def sample_user_negatives(self, U, I, train_ptr, train_idx, test_ptr, test_idx, n, seed=0):
        """dataset.SampleUserNegatives (dataset.go:242-253): (neg U x n int32 padded with -1, len U)"""
        train_ptr, p1 = _i64(train_ptr)
        srt, p2 = _i32(sort_rows(train_ptr, np.asarray(train_idx, np.int32)))
        test_ptr, p3 = _i64(test_ptr)
        test_idx, p4 = _i32(test_idx if len(test_idx) else np.zeros(1, np.int32))
        out = np.zeros((U, n), np.int32)
        ln = np.zeros(U, np.int32)
        self.L.orc_sample_user_negatives(U, I, p1, p2, p3, p4, n, seed, out.ctypes.data_as(_i32p), ln.ctypes.data_as(_i32p))
        return out, ln

    # ---- BPR / ALS -------------------------------------------------------
    def bpr_apply_triplets(self, P, Q, u, i, j, lr, reg):
        """Sequential (Jobs=1) SGD over the triplet stream; returns new (P, Q, cost)."""
        P = np.array(P, dtype=np.float32, order="C")
        Q = np.array(Q, dtype=np.float32, order="C")
        u, pu = _i32(u)
        i, pi = _i32(i)
        j, pj = _i32(j)
        cost = self.L.orc_bpr_apply_triplets(P.ctypes.data_as(_f32p), Q.ctypes.data_as(_f32p), P.shape[1], pu, pi, pj,
                                             u.size, lr, reg)
        return P, Q, cost

    def bpr_epoch_sampled(self, P, Q, indptr, indices, sorted_indices, seed, epoch, sample_base, n, lr, reg):
        """In-place epoch (P, Q must be C-contiguous float32); used by tests and the cpu_baseline leg."""
        assert P.dtype == np.float32 and Q.dtype == np.float32 and P.flags.c_contiguous and Q.flags.c_contiguous
        indptr, pp = _i64(indptr)
        indices, pi = _i32(indices)
        sorted_indices, ps = _i32(sorted_indices)
        return self.L.orc_bpr_epoch_sampled(P.ctypes.data_as(_f32p), Q.ctypes.data_as(_f32p), P.shape[0], Q.shape[0],
                                            P.shape[1], pp, pi, ps, seed, epoch, sample_base, n, lr, reg)

    def als_epoch(self, P, Q, uptr, uidx, iptr, iidx, w, reg):
        P = np.array(P, dtype=np.float32, order="C")
        Q = np.array(Q, dtype=np.float32, order="C")
        uptr, a = _i64(uptr)
        uidx, b = _i32(uidx)
        iptr, c = _i64(iptr)
        iidx, d = _i32(iidx)
        self.L.orc_als_epoch(P.ctypes.data_as(_f32p), Q.ctypes.data_as(_f32p), P.shape[0], Q.shape[0], P.shape[1], a,
                             b, c, d, w, reg)
        return P, Q

    def als_half_range(self, A, B, ptr, idx, bptr, w, reg, row_begin, row_end):
        """rows [row_begin, row_end) of A solved in place against B (model.go:659-690 / 707-738)"""
        assert A.dtype == np.float32 and A.flags.c_contiguous
        B, pB = _f32(B)
        ptr, a = _i64(ptr)
        idx, b = _i32(idx)
        bptr, c = _i64(bptr)
        self.L.orc_als_half_range(A.ctypes.data_as(_f32p), pB, A.shape[0], B.shape[0], A.shape[1], a, b, c, w, reg,
                                  row_begin, row_end)
        return A

    # ---- predict / evaluate ------------------------------------------------
    def mf_score(self, P, Q, u, i):
        P, pP = _f32(P)
        Q, pQ = _f32(Q)
        u, pu = _i32(u)
        i, pi = _i32(i)
        out = np.zeros(u.size, dtype=np.float32)
        self.L.orc_mf_score(pP, pQ, P.shape[1], pu, pi, u.size, out.ctypes.data_as(_f32p))
        return out

    def mf_rank(self, P, Q, users, cand_ptr, cand, topk):
        P, pP = _f32(P)
        Q, pQ = _f32(Q)
        users, pu = _i32(users)
        cand_ptr, pp = _i64(cand_ptr)
        cand, pc = _i32(cand)
        rank = np.full((users.size, topk), -1, dtype=np.int32)
        rlen = np.zeros(users.size, dtype=np.int32)
        self.L.orc_mf_rank(pP, pQ, P.shape[1], users.size, pu, pp, pc, topk, rank.ctypes.data_as(_i32p),
                           rlen.ctypes.data_as(_i32p))
        return rank, rlen

    def metric(self, m, target, rank):
        target, pt = _i32(target)
        rank, pr = _i32(rank)
        return float(np.float32(self.L.orc_metric(m, pt, target.size, pr, rank.size)))

    def evaluate(self, P, Q, test_ptr, test_idx, neg_ptr, neg_idx, topk, metrics=(M_NDCG, M_PRECISION, M_RECALL)):
        P, pP = _f32(P)
        Q, pQ = _f32(Q)
        test_ptr, a = _i64(test_ptr)
        test_idx, b = _i32(test_idx)
        neg_ptr, c = _i64(neg_ptr)
        neg_idx, d = _i32(neg_idx)
        metrics, pm = _i32(list(metrics))
        out = np.zeros(metrics.size, dtype=np.float32)
        self.L.orc_evaluate(pP, pQ, P.shape[1], P.shape[0], a, b, c, d, topk, pm, metrics.size,
                            out.ctypes.data_as(_f32p))
        return out


def sort_rows(indptr, indices):
    """Each CSR row sorted ascending (membership structure of the BPR negative sampler)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices, dtype=np.int32)
    rows = np.repeat(np.arange(indptr.size - 1, dtype=np.int64), np.diff(indptr))
    order = np.lexsort((indices, rows))
    return np.ascontiguousarray(indices[order])


class Ref:
    """The reference's own C kernels (oracle/_ref/libgorse_ref.so)."""

    def __init__(self, path):
        L = C.CDLL(path)
        self.L = L
        for isa in ("_mm512", "_mm256"):
            getattr(L, isa + "_dot").restype = C.c_float
            getattr(L, isa + "_dot").argtypes = [_f32p, _f32p, C.c_int64]
            getattr(L, isa + "_euclidean").restype = C.c_float
            getattr(L, isa + "_euclidean").argtypes = [_f32p, _f32p, C.c_int64]
            getattr(L, isa + "_euclidean_bf16").restype = C.c_float
            getattr(L, isa + "_euclidean_bf16").argtypes = [_u16p, _u16p, C.c_int64]
            getattr(L, isa + "_mul_const_add").argtypes = [_f32p, _f32p, _f32p, C.c_int64]
            getattr(L, isa + "_mul_const_add_to").argtypes = [_f32p, _f32p, _f32p, _f32p, C.c_int64]
            getattr(L, isa + "_mul_const_to").argtypes = [_f32p, _f32p, _f32p, C.c_int64]
            getattr(L, isa + "_sub_to").argtypes = [_f32p, _f32p, _f32p, C.c_int64]
            getattr(L, isa + "_mm").argtypes = [C.c_bool, C.c_bool, C.c_int64, C.c_int64, C.c_int64, _f32p, C.c_int64,
                                                _f32p, C.c_int64, _f32p, C.c_int64]

    @staticmethod
    def _p(isa):
        return "_mm512" if isa == ISA_AVX512 else "_mm256"

    def dot(self, isa, a, b):
        a, pa = _f32(a)
        b, pb = _f32(b)
        return float(np.float32(getattr(self.L, self._p(isa) + "_dot")(pa, pb, a.size)))

    def euclidean(self, isa, a, b):
        a, pa = _f32(a)
        b, pb = _f32(b)
        return float(np.float32(getattr(self.L, self._p(isa) + "_euclidean")(pa, pb, a.size)))

    def euclidean_bf16(self, isa, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint16)
        b = np.ascontiguousarray(b, dtype=np.uint16)
        return float(np.float32(getattr(self.L, self._p(isa) + "_euclidean_bf16")(
            a.ctypes.data_as(_u16p), b.ctypes.data_as(_u16p), a.size)))

    def mul_const_add(self, isa, a, c, dst):
        a, pa = _f32(a)
        dst = np.array(dst, dtype=np.float32)
        cc = C.c_float(c)
        getattr(self.L, self._p(isa) + "_mul_const_add")(pa, C.cast(C.byref(cc), _f32p), dst.ctypes.data_as(_f32p),
                                                         a.size)
        return dst

    def mul_const_add_to(self, isa, a, b, c):
        a, pa = _f32(a)
        c, pc = _f32(c)
        dst = np.zeros_like(a)
        bb = C.c_float(b)
        getattr(self.L, self._p(isa) + "_mul_const_add_to")(pa, C.cast(C.byref(bb), _f32p), pc,
                                                            dst.ctypes.data_as(_f32p), a.size)
        return dst

    def mm(self, isa, transA, transB, m, n, k, a, lda, b, ldb, c, ldc):
        a, pa = _f32(a)
        b, pb = _f32(b)
        c = np.array(c, dtype=np.float32)
        getattr(self.L, self._p(isa) + "_mm")(bool(transA), bool(transB), m, n, k, pa, lda, pb, ldb,
                                              c.ctypes.data_as(_f32p), ldc)
        return c


def _cpu_has(flag):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return flag in line.split()
    except OSError:
        pass
    return False


def load_ref():
    """The compiled reference kernels, or None (file absent / host lacks AVX512F+FMA)."""
    path = os.path.join(HERE, "_ref", "libgorse_ref.so")
    if not os.path.exists(path):
        return None
    if not (_cpu_has("avx512f") and _cpu_has("avx512bw") and _cpu_has("fma") and _cpu_has("avx2")):
        return None
    return Ref(path)


# ---- ann.HNSW search restated (test infrastructure, pure Python: small cases only) ---------------------------------------
def hnsw_parse_index_section(blob, at):
    """HNSW.Unmarshal (common/ann/hnsw.go:339-418) over the bytes of an index section starting at `at`: returns
    (params dict, vector gob streams, bottom = list of [(value, weight)], upper = list of {key: [(value, weight)]}, enterPoint,
    position behind the section).  A queue = PriorityQueue.Marshal (common/heap/pq.go:128-133): bool desc, int32 length,
    (int32 value, float32 weight) pairs."""
    import struct

    def queue():
        nonlocal at
        desc, ln = struct.unpack_from("<?i", blob, at)
        at += 5
        q = [struct.unpack_from("<if", blob, at + 8 * j) for j in range(ln)]
        at += 8 * ln
        return desc, q
    level_factor, m, m0, ef, efc = struct.unpack_from("<fqqqq", blob, at)
    at += 36
    (n,) = struct.unpack_from("<q", blob, at)
    at += 8
    streams = []
    for _ in range(n):
        (ln,) = struct.unpack_from("<i", blob, at)
        streams.append(blob[at + 4:at + 4 + ln])
        at += 4 + ln
    bottom = [queue() for _ in range(n)]
    (layers,) = struct.unpack_from("<q", blob, at)
    at += 8
    upper = []
    for _ in range(layers):
        (cnt,) = struct.unpack_from("<i", blob, at)
        at += 4
        layer = {}
        for _ in range(cnt):
            (key,) = struct.unpack_from("<i", blob, at)
            at += 4
            layer[key] = queue()
        upper.append(layer)
    (enter,) = struct.unpack_from("<i", blob, at)
    at += 4
    params = {"levelFactor": level_factor, "maxConnection": m, "maxConnection0": m0, "ef": ef, "efConstruction": efc}
    return params, streams, bottom, upper, enter, at


def hnsw_knn_search(X, bottom, upper, enter, params, q, k):
    """HNSW.SearchVector (hnsw.go:88-98) = knnSearch (hnsw.go:100-114) over searchLayer (hnsw.go:187-229) and selectNeighbors
    (hnsw.go:254-260) with distance = -floats.Dot (logics/cf.go:32-34), on a graph given as parsed queues.  The two heaps of
    searchLayer are Python heapq's (container/heap's sift rules matter only between EQUAL distances, which a recall check does
    not depend on).  Returns the ids, nearest first."""
    import heapq

    def dist(i):
        return float(-np.dot(X[i].astype(np.float32), q.astype(np.float32)))

    def neighbours(c, layer):
        return [v for v, _ in (bottom[c][1] if layer == 0 else upper[layer - 1][c][1])]

    def search_layer(enter_points, ef, layer):
        visited = set(i for _, i in enter_points)
        cand = list(enter_points)  # min-heap by distance
        heapq.heapify(cand)
        w = [(-dd, i) for dd, i in enter_points]  # max-heap by distance
        heapq.heapify(w)
        while cand:
            cq, c = heapq.heappop(cand)
            if cq > -w[0][0]:
                break
            for e in neighbours(c, layer):
                if e in visited:
                    continue
                visited.add(e)
                eq = dist(e)
                if eq < -w[0][0] or len(w) < ef:
                    heapq.heappush(cand, (eq, e))
                    heapq.heappush(w, (-eq, e))
                    if len(w) > ef:
                        heapq.heappop(w)
        return sorted((-nd, i) for nd, i in w)
    ef = max(params["ef"], k) if params["ef"] > 0 else max(params["efConstruction"], k)  # efSearchValue, hnsw.go:268-273
    points = [(dist(enter), enter)]
    for layer in range(len(upper), 0, -1):
        w = search_layer(points, 1, layer)
        points = [w[0]]
    w = search_layer(points, ef, 0)
    return [i for _, i in w[:k]]
