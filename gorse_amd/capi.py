"""ctypes binding of the C ABI in include/gorse_hip.h (libgorse_hip.so).

This is plumbing: every entry point of the header is declared here with its exact
signature, plus thin numpy-friendly wrappers (`MF`, `TopK`, `Sparse`).  There is NO CPU fallback:
if the shared library is missing, or no gfx950 device is visible when a handle is
created, the call raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GORSE_HIP_LIB") or os.path.join(HERE, "lib", "libgorse_hip.so")  # the override serves probe builds

OK, ERR_INVALID, ERR_HIP, ERR_CANCELLED, ERR_NO_DEVICE, ERR_RANGE, ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6
BPR_HOGWILD_ATOMIC, BPR_SEQUENTIAL, BPR_HOGWILD_RACY, BPR_HOGWILD_STORES = 0, 1, 2, 3
DTYPE_F32, DTYPE_BF16 = 0, 1
METRIC_NEG_DOT, METRIC_EUCLIDEAN, METRIC_COSINE, METRIC_EUCLIDEAN_BF16 = 0, 1, 2, 3
PROF_BPR_UPDATE, PROF_BPR_SAMPLE, PROF_ALS_SWEEP, PROF_ALS_GRAM, PROF_BPR_SORT, PROF_COMM = 0, 1, 2, 3, 4, 5
COMM_ID_BYTES = 128
PROF_TOPK_SCORE, PROF_TOPK_RESCORE, PROF_TOPK_SWEEP, PROF_TOPK_SELECT, PROF_TOPK_HIST, PROF_TOPK_REPLAY = 0, 1, 2, 3, 4, 5

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/gorse_hip.h (the boundary) and include/gorse_hip_test.h (test hooks) one to one
SIGNATURES = {
    "gorse_hip_abi_version": (C.c_int32, []),
    "gorse_hip_last_error": (C.c_char_p, []),
    "gorse_hip_device_count": (C.c_int32, [_i32p]),
    "gorse_mf_create": (C.c_int32, [C.POINTER(_vp), C.c_int32, C.c_int64, C.c_int64, C.c_int32, _i64p, _i32p, _i64p, _i32p]),
    "gorse_mf_destroy": (C.c_int32, [_vp]),
    "gorse_mf_set_factors": (C.c_int32, [_vp, _f32p, _f32p]),
    "gorse_mf_get_factors": (C.c_int32, [_vp, _f32p, _f32p]),
    "gorse_mf_score": (C.c_int32, [_vp, _i32p, _i32p, C.c_int64, _f32p]),
    "gorse_mf_rank": (C.c_int32, [_vp, C.c_int64, _i32p, _i64p, _i32p, C.c_int32, _i32p, _i32p]),
    "gorse_mf_sample_user_negatives": (C.c_int32, [_vp, _i64p, _i32p, C.c_int32, C.c_uint64, _i32p, _i32p]),
    "gorse_mf_resident_candidates": (C.c_int32, [_vp, _i64p, _i64p]),
    "gorse_mf_resident_generation": (C.c_int32, [_vp, C.POINTER(C.c_uint64)]),
    "gorse_mf_rank_resident": (C.c_int32, [_vp, C.c_int32, _i32p, _i32p, _i32p]),
    "gorse_bpr_epoch": (C.c_int32, [_vp, C.c_int64, C.c_float, C.c_float, C.c_uint64, C.c_uint64, C.c_int64, C.c_int32,
                                    _i32p, _f64p]),
    "gorse_bpr_epoch_enqueue": (C.c_int32, [_vp, C.c_int64, C.c_float, C.c_float, C.c_uint64, C.c_uint64, C.c_int64,
                                            C.c_int32]),
    "gorse_bpr_sample_triplets": (C.c_int32, [_vp, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, _i32p, _i32p, _i32p]),
    "gorse_bpr_apply_triplets": (C.c_int32, [_vp, _i32p, _i32p, _i32p, C.c_int64, C.c_float, C.c_float, C.c_int32]),
    "gorse_mf_bpr_schedule": (C.c_int32, [_vp, _i32p]),
    "gorse_mf_set_bpr_cold_window": (C.c_int32, [_vp, C.c_int64, _i64p]),
    "gorse_als_epoch": (C.c_int32, [_vp, C.c_float, C.c_float, _i32p]),
    "gorse_als_set_ranges": (C.c_int32, [_vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "gorse_als_half_epoch": (C.c_int32, [_vp, C.c_int32, C.c_float, C.c_float]),
    "gorse_als_half_epoch_enqueue": (C.c_int32, [_vp, C.c_int32, C.c_float, C.c_float]),
    "gorse_mf_rows_export": (C.c_int32, [_vp, C.c_int32, C.c_int64, C.c_int64, _vp]),
    "gorse_mf_rows_import": (C.c_int32, [_vp, C.c_int32, C.c_int64, C.c_int64, _vp]),
    "gorse_mf_item_sync_mark": (C.c_int32, [_vp]),
    "gorse_mf_item_delta_export": (C.c_int32, [_vp, _vp]),
    "gorse_mf_item_delta_import": (C.c_int32, [_vp, _vp]),
    "gorse_mf_device_ptrs": (C.c_int32, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "gorse_comm_unique_id": (C.c_int32, [_vp]),
    "gorse_comm_create": (C.c_int32, [C.POINTER(_vp), _vp, C.c_int32, C.c_int32, C.c_int32]),
    "gorse_comm_create_local": (C.c_int32, [C.POINTER(_vp), _i32p, C.c_int32]),
    "gorse_comm_destroy": (C.c_int32, [_vp]),
    "gorse_comm_info": (C.c_int32, [_vp, _i32p, _i32p]),
    "gorse_mf_item_allreduce": (C.c_int32, [C.POINTER(_vp), C.POINTER(_vp), C.c_int32]),
    "gorse_mf_rows_allgather": (C.c_int32, [C.POINTER(_vp), C.POINTER(_vp), C.c_int32, C.c_int32, _i64p]),
    "gorse_comm_allreduce_f32": (C.c_int32, [_vp, _f32p, C.c_int64]),
    "gorse_comm_allreduce_f32_local": (C.c_int32, [C.POINTER(_vp), C.c_int32, C.POINTER(_f32p), C.c_int64]),
    "gorse_comm_available": (C.c_int32, []),
    "gorse_mf_synchronize": (C.c_int32, [_vp]),
    "gorse_mf_epoch_throttle": (C.c_int32, [_vp, C.c_int32, _i32p]),
    "gorse_mf_epoch_times": (C.c_int32, [_vp, _i64p, C.POINTER(C.c_double), _i64p, C.c_int32]),
    "gorse_mf_set_profiling": (C.c_int32, [_vp, C.c_int32]),
    "gorse_mf_get_profile": (C.c_int32, [_vp, C.c_int32, _i64p, _f64p]),
    "gorse_mf_reset_profile": (C.c_int32, [_vp]),
    "gorse_topk_create": (C.c_int32, [C.POINTER(_vp), C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "gorse_topk_destroy": (C.c_int32, [_vp]),
    "gorse_topk_search_index": (C.c_int32, [_vp, _i64p, C.c_int64, C.c_int32, C.c_int32, _i32p, _f32p, _i32p]),
    "gorse_topk_search_vector": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int32, C.c_int32, _i32p, _f32p, _i32p]),
    "gorse_topk_all_pairs": (C.c_int32, [_vp, C.c_int64, C.c_int64, C.c_int32, _i32p, _f32p]),
    "gorse_topk_set_mask": (C.c_int32, [_vp, _vp]),
    "gorse_topk_synchronize": (C.c_int32, [_vp]),
    "gorse_topk_set_profiling": (C.c_int32, [_vp, C.c_int32]),
    "gorse_topk_get_profile": (C.c_int32, [_vp, C.c_int32, _i64p, _f64p]),
    "gorse_topk_last_stats": (C.c_int32, [_vp, _i64p, _i64p]),
    "gorse_sparse_create": (C.c_int32, [C.POINTER(_vp), C.c_int32, C.c_int64, _i64p, _vp, _f32p]),
    "gorse_sparse_destroy": (C.c_int32, [_vp]),
    "gorse_sparse_set_mask": (C.c_int32, [_vp, _vp]),
    "gorse_sparse_search": (C.c_int32, [_vp, C.c_int64, _i64p, _vp, _f32p, _i64p, C.c_int32, _i32p, _f32p, _i32p]),
    "gorse_sparse_all_pairs": (C.c_int32, [_vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, _i32p, _f32p, _i32p]),
    "gorse_sparse_synchronize": (C.c_int32, [_vp]),
    "gorse_sparse_set_profiling": (C.c_int32, [_vp, C.c_int32]),
    "gorse_sparse_get_profile": (C.c_int32, [_vp, _i64p, _f64p]),
    "gorse_sparse_last_stats": (C.c_int32, [_vp, _i64p, _i64p]),
    "gorse_hip_sgemm": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p, C.c_int32,
                                    _f32p, C.c_int32, _f32p, C.c_int32]),
    "gorse_hip_sgemm_device": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32,
                                           _vp, C.c_int32, _vp, C.c_int32]),
    "gorse_hip_test_set_exact_exp": (None, [C.c_int32]),
    "gorse_hip_test_set_variant": (None, [C.c_int32]),
    "gorse_hip_test_set_topk_path": (None, [C.c_int32]),
    "gorse_hip_test_set_topk_variant": (None, [C.c_int32]),
    "gorse_hip_test_get_sweep_profile": (C.c_int32, [_vp, C.POINTER(C.c_uint64)]),
    "gorse_topk_tri_begin": (C.c_int32, [_vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "gorse_topk_tri_slice": (C.c_int32, [_vp, C.c_int32, _i64p, _i64p, _i64p]),
    "gorse_topk_tri_thresholds_get": (C.c_int32, [_vp, C.c_int64, C.c_int64, _vp]),
    "gorse_topk_tri_thresholds_put": (C.c_int32, [_vp, C.c_int64, C.c_int64, _vp]),
    "gorse_topk_tri_sweep": (C.c_int32, [_vp]),
    "gorse_topk_tri_pack": (C.c_int32, [_vp, C.c_int32, _i64p, _i64p]),
    "gorse_topk_tri_pack_read": (C.c_int32, [_vp, _vp, _vp]),
    "gorse_topk_tri_unpack": (C.c_int32, [_vp, C.c_int32, _vp, C.c_int64, _vp, C.c_int64]),
    "gorse_topk_tri_finish": (C.c_int32, [_vp, _i32p, _f32p]),
    "gorse_topk_tri_all_pairs_local": (C.c_int32, [C.POINTER(_vp), C.c_int32, C.c_int64, C.c_int64, C.c_int32, _i32p, _f32p]),
    "gorse_hip_test_topk_resweeps": (C.c_int32, [_vp, _i64p]),
    "gorse_hip_test_topk_last_symmetric": (C.c_int32, [_vp, C.POINTER(C.c_int32)]),
    "gorse_hip_test_topk_sym_stats": (C.c_int32, [_vp, C.POINTER(C.c_uint64)]),
    "gorse_hip_test_topk_get_thresholds": (C.c_int32, [_vp, _f32p, C.c_int64]),
    "gorse_hip_test_topk_get_pilot_state": (C.c_int32, [_vp, C.POINTER(C.c_uint8), _i32p, C.c_int64]),
    "gorse_hip_test_topk_get_flags": (C.c_int32, [_vp, C.POINTER(C.c_uint8), C.c_int64]),
    "gorse_hip_test_topk_get_foreign_counts": (C.c_int32, [_vp, _i32p, C.c_int64]),
    "gorse_hip_test_set_sparse_slots": (None, [C.c_int64]),
    "gorse_hip_test_set_sparse_head": (None, [C.c_int32]),
    "gorse_hip_test_set_sparse_table": (None, [C.c_int32]),
    "gorse_hip_test_set_sparse_probe": (None, [C.c_int32]),
    "gorse_hip_test_set_sparse_sym": (None, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "gorse_hip_test_set_sparse_front": (None, [C.c_int32]),
    "gorse_hip_test_sparse_sym_stats": (None, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gorse_hip_test_set_sparse_tile": (None, [C.c_int32]),
    "gorse_hip_test_set_scan_literal": (None, [C.c_int32]),
    "gorse_hip_test_set_stream_priorities": (None, [C.c_int32]),
    "gorse_hip_test_set_sparse_split": (None, [C.c_int64]),
    "gorse_hip_test_set_sparse_heavy": (None, [C.c_int64]),
    "gorse_hip_test_set_sparse_atomic": (None, [C.c_int32]),
    "gorse_hip_test_sparse_trace": (C.c_int64, [_vp, C.c_int32, C.POINTER(C.c_uint64), C.c_int64]),
    "gorse_hip_test_set_als_path": (None, [C.c_int32]),
    "gorse_hip_test_set_als_plan": (None, [C.c_int32, C.c_int32]),
    "gorse_hip_test_als_profile": (C.c_int32, [_vp, C.c_int32, C.POINTER(C.c_uint64)]),
    "gorse_hip_test_set_bpr_chunk": (None, [C.c_int64]),
    "gorse_hip_test_bpr_prepare_chunk": (C.c_int32, [_vp, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, _i32p, _i32p, _i32p]),
    "gorse_hip_test_set_bpr_store_mode": (None, [C.c_int32]),
    "gorse_hip_test_set_prep_cu_stride": (None, [C.c_int32]),
    "gorse_hip_test_set_bpr_user_segments": (None, [C.c_int32]),
    "gorse_hip_test_set_bpr_user_depth": (None, [C.c_int32]),
    "gorse_hip_test_probe_build": (C.c_int32, []),
    "gorse_hip_test_set_sgemm_valu": (None, [C.c_int32]),
    "gorse_hip_test_sgemm_last_ms": (C.c_double, []),
    "gorse_hip_test_set_bpr_cold_window": (None, [C.c_int64]),
}


class GorseHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libgorse_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load libgorse_hip.so (once). Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP extension is mandatory; there is no CPU path)" % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise GorseHipError(rc, lib().gorse_hip_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int32(0)
    rc = lib().gorse_hip_device_count(C.byref(n))
    return n.value if rc == OK else 0


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class MF:
    """One gorse_mf handle (one model resident on one GPU)."""

    def __init__(self, U, I, d, user_indptr, user_indices, item_indptr=None, item_indices=None, device=0):
        self.U, self.I, self.d = int(U), int(I), int(d)
        up, ui = _arr(user_indptr, np.int64), _arr(user_indices, np.int32)
        if up.size != self.U + 1:
            raise GorseHipError(ERR_INVALID, "user_indptr must have U+1 entries")
        ip = _arr(item_indptr, np.int64) if item_indptr is not None else None
        ii = _arr(item_indices, np.int32) if item_indices is not None else None
        self.h = _vp()
        check(lib().gorse_mf_create(C.byref(self.h), device, self.U, self.I, self.d, _p(up, _i64p), _p(ui, _i32p),
                                    _p(ip, _i64p), _p(ii, _i32p)))

    def close(self):
        if getattr(self, "h", None):
            lib().gorse_mf_destroy(self.h)
            self.h = None

    __del__ = close

    def set_factors(self, P=None, Q=None):
        P = _arr(P, np.float32) if P is not None else None
        Q = _arr(Q, np.float32) if Q is not None else None
        if P is not None and P.shape != (self.U, self.d):
            raise GorseHipError(ERR_INVALID, "P must be U x d")
        if Q is not None and Q.shape != (self.I, self.d):
            raise GorseHipError(ERR_INVALID, "Q must be I x d")
        check(lib().gorse_mf_set_factors(self.h, _p(P, _f32p), _p(Q, _f32p)))

    def get_factors(self):
        P = np.empty((self.U, self.d), np.float32)
        Q = np.empty((self.I, self.d), np.float32)
        check(lib().gorse_mf_get_factors(self.h, _p(P, _f32p), _p(Q, _f32p)))
        return P, Q

    def score(self, u, i):
        u, i = _arr(u, np.int32), _arr(i, np.int32)
        out = np.empty(u.size, np.float32)
        check(lib().gorse_mf_score(self.h, _p(u, _i32p), _p(i, _i32p), u.size, _p(out, _f32p)))
        return out

    def rank(self, users, cand_indptr, cand, topk):
        users, cand_indptr, cand = _arr(users, np.int32), _arr(cand_indptr, np.int64), _arr(cand, np.int32)
        rank = np.empty((users.size, topk), np.int32)
        rlen = np.empty(users.size, np.int32)
        check(lib().gorse_mf_rank(self.h, users.size, _p(users, _i32p), _p(cand_indptr, _i64p), _p(cand, _i32p), topk,
                                  _p(rank, _i32p), _p(rlen, _i32p)))
        return rank, rlen

    def sample_user_negatives(self, test_indptr, test_indices, num_candidates, seed=0, fetch=True):
        """dataset.SampleUserNegatives on the device: (neg U x n padded with -1, len U); the candidate lists of the users with
        test feedback stay resident for rank_resident"""
        tp, ti = _arr(test_indptr, np.int64), _arr(test_indices, np.int32)
        if ti.size == 0:
            ti = np.zeros(1, np.int32)
        neg = np.empty((self.U, num_candidates), np.int32) if fetch else None
        ln = np.empty(self.U, np.int32) if fetch else None
        check(lib().gorse_mf_sample_user_negatives(self.h, _p(tp, _i64p), _p(ti, _i32p), num_candidates, seed,
                                                   _p(neg, _i32p) if fetch else None, _p(ln, _i32p) if fetch else None))
        return neg, ln

    def rank_resident(self, topk):
        """Rank over the resident candidate lists: (users, rank n_users x topk padded with -1, lengths)"""
        nu, nc = C.c_int64(0), C.c_int64(0)
        check(lib().gorse_mf_resident_candidates(self.h, C.byref(nu), C.byref(nc)))
        users = np.empty(nu.value, np.int32)
        rank = np.empty((nu.value, topk), np.int32)
        rlen = np.empty(nu.value, np.int32)
        if nu.value:
            check(lib().gorse_mf_rank_resident(self.h, topk, _p(users, _i32p), _p(rank, _i32p), _p(rlen, _i32p)))
        return users, rank, rlen

    def set_bpr_cold_window(self, samples):
        """the cold window of THIS handle (GORSE_BPR_HOGWILD_STORES); returns the number of cold items"""
        n = C.c_int64(0)
        check(lib().gorse_mf_set_bpr_cold_window(self.h, samples, C.byref(n)))
        return n.value

    def bpr_epoch(self, n_samples, lr, reg, seed, epoch, sample_base=0, mode=BPR_HOGWILD_STORES, want_loss=False,
                  cancel=None):
        loss = C.c_double(0)
        cp = _p(cancel, _i32p) if cancel is not None else None
        check(lib().gorse_bpr_epoch(self.h, n_samples, lr, reg, seed, epoch, sample_base, mode, cp,
                                    C.byref(loss) if want_loss else None))
        return loss.value

    def bpr_epoch_enqueue(self, n_samples, lr, reg, seed, epoch, sample_base=0, mode=BPR_HOGWILD_STORES):
        check(lib().gorse_bpr_epoch_enqueue(self.h, n_samples, lr, reg, seed, epoch, sample_base, mode))

    def bpr_sample_triplets(self, n, seed, epoch, sample_base=0):
        u = np.empty(n, np.int32)
        i = np.empty(n, np.int32)
        j = np.empty(n, np.int32)
        check(lib().gorse_bpr_sample_triplets(self.h, n, seed, epoch, sample_base, _p(u, _i32p), _p(i, _i32p),
                                              _p(j, _i32p)))
        return u, i, j

    def bpr_prepare_chunk(self, n, seed, epoch, sample_base=0):
        """test hook: (run offsets U + 2, positives n, negatives n) of the user-run schedule's preparation of one chunk"""
        off = np.empty(self.U + 2, np.int32)
        si = np.empty(n, np.int32)
        sj = np.empty(n, np.int32)
        check(lib().gorse_hip_test_bpr_prepare_chunk(self.h, n, seed, epoch, sample_base, _p(off, _i32p), _p(si, _i32p),
                                                     _p(sj, _i32p)))
        return off, si, sj

    def bpr_apply_triplets(self, u, i, j, lr, reg, mode):
        u, i, j = _arr(u, np.int32), _arr(i, np.int32), _arr(j, np.int32)
        if not (u.size == i.size == j.size):
            raise GorseHipError(ERR_INVALID, "triplet arrays differ in length")
        check(lib().gorse_bpr_apply_triplets(self.h, _p(u, _i32p), _p(i, _i32p), _p(j, _i32p), u.size, lr, reg, mode))

    def als_epoch(self, weight, reg, cancel=None):
        cp = _p(cancel, _i32p) if cancel is not None else None
        check(lib().gorse_als_epoch(self.h, weight, reg, cp))

    def item_sync_mark(self):
        check(lib().gorse_mf_item_sync_mark(self.h))

    def bpr_user_runs(self):
        v = C.c_int32(0)
        check(lib().gorse_mf_bpr_schedule(self.h, C.byref(v)))
        return bool(v.value)

    def als_profile(self, enable, fetch=False):
        out = (C.c_uint64 * 16)() if fetch else None
        check(lib().gorse_hip_test_als_profile(self.h, int(enable), out))
        return [int(x) for x in out] if fetch else None

    def als_set_ranges(self, u_begin, u_end, i_begin, i_end):
        check(lib().gorse_als_set_ranges(self.h, u_begin, u_end, i_begin, i_end))

    def als_half_epoch(self, side, weight, reg):
        check(lib().gorse_als_half_epoch(self.h, side, weight, reg))

    def als_half_epoch_enqueue(self, side, weight, reg):
        check(lib().gorse_als_half_epoch_enqueue(self.h, side, weight, reg))

    def rows_export(self, side, begin, end, dev_ptr):
        check(lib().gorse_mf_rows_export(self.h, side, begin, end, _vp(dev_ptr)))

    def rows_import(self, side, begin, end, dev_ptr):
        check(lib().gorse_mf_rows_import(self.h, side, begin, end, _vp(dev_ptr)))

    def item_delta_export(self, dev_ptr):
        check(lib().gorse_mf_item_delta_export(self.h, _vp(dev_ptr)))

    def item_delta_import(self, dev_ptr):
        check(lib().gorse_mf_item_delta_import(self.h, _vp(dev_ptr)))

    def device_ptrs(self):
        P, Q = _vp(), _vp()
        check(lib().gorse_mf_device_ptrs(self.h, C.byref(P), C.byref(Q)))
        return P.value, Q.value

    def synchronize(self):
        check(lib().gorse_mf_synchronize(self.h))

    def epoch_throttle(self, max_in_flight, cancel=None):
        """blocks until at most max_in_flight enqueued epochs are unfinished; raises GorseHipError(ERR_CANCELLED) when *cancel is set"""
        check(lib().gorse_mf_epoch_throttle(self.h, max_in_flight, _p(cancel, _i32p) if cancel is not None else None))

    def epoch_times(self, reset=False):
        """(finished epochs, their device milliseconds, epochs still in flight) since the last reset"""
        n, ms, fl = C.c_int64(0), C.c_double(0), C.c_int64(0)
        check(lib().gorse_mf_epoch_times(self.h, C.byref(n), C.byref(ms), C.byref(fl), int(bool(reset))))
        return n.value, ms.value, fl.value

    def set_profiling(self, on):
        check(lib().gorse_mf_set_profiling(self.h, int(bool(on))))

    def reset_profile(self):
        check(lib().gorse_mf_reset_profile(self.h))

    def get_profile(self, cls):
        n, ms = C.c_int64(0), C.c_double(0)
        check(lib().gorse_mf_get_profile(self.h, cls, C.byref(n), C.byref(ms)))
        return n.value, ms.value


class TopK:
    """One gorse_topk handle (exact brute-force index resident on one GPU)."""

    def __init__(self, X, metric, dtype=DTYPE_F32, device=0):
        X = _arr(X, np.uint16 if dtype == DTYPE_BF16 else np.float32)
        if X.ndim != 2:
            raise GorseHipError(ERR_INVALID, "X must be N x d")
        self.N, self.d = X.shape
        self.dtype, self.metric = dtype, metric
        self.h = _vp()
        check(lib().gorse_topk_create(C.byref(self.h), device, self.N, self.d, dtype, metric, X.ctypes.data_as(_vp)))

    def close(self):
        if getattr(self, "h", None):
            lib().gorse_topk_destroy(self.h)
            self.h = None

    __del__ = close

    def _out(self, nq, k):
        return np.empty((nq, k), np.int32), np.empty((nq, k), np.float32), np.empty(nq, np.int32)

    def search_index(self, q, k, prune0=False):
        q = _arr(np.atleast_1d(q), np.int64)
        idx, dist, cnt = self._out(q.size, k)
        check(lib().gorse_topk_search_index(self.h, _p(q, _i64p), q.size, k, int(prune0), _p(idx, _i32p),
                                            _p(dist, _f32p), _p(cnt, _i32p)))
        return idx, dist, cnt

    def search_vector(self, qv, k, prune0=False):
        qv = _arr(np.atleast_2d(qv), np.uint16 if self.dtype == DTYPE_BF16 else np.float32)
        if qv.shape[1] != self.d:
            raise GorseHipError(ERR_INVALID, "query dimension mismatch")
        idx, dist, cnt = self._out(qv.shape[0], k)
        check(lib().gorse_topk_search_vector(self.h, qv.ctypes.data_as(_vp), qv.shape[0], k, int(prune0),
                                             _p(idx, _i32p), _p(dist, _f32p), _p(cnt, _i32p)))
        return idx, dist, cnt

    def all_pairs(self, k, q_begin=0, q_end=None, fetch=True):
        q_end = self.N if q_end is None else q_end
        nq = q_end - q_begin
        idx = np.empty((nq, k), np.int32) if fetch else None
        dist = np.empty((nq, k), np.float32) if fetch else None
        check(lib().gorse_topk_all_pairs(self.h, q_begin, q_end, k, _p(idx, _i32p), _p(dist, _f32p)))
        return idx, dist

    # ---- the triangle-sharded all-pairs search (include/gorse_hip.h, gorse_topk_tri_*): host-memory forms of the messages ----
    def tri_begin(self, k, rank, world, q_begin=0, q_end=None):
        q_end = self.N if q_end is None else q_end
        self._tri = (k, q_end - q_begin)
        check(lib().gorse_topk_tri_begin(self.h, q_begin, q_end, k, rank, world))

    def tri_slice(self, rank):
        """(lo, hi, owned): the queries whose pilots `rank` runs, and how many queries it owns"""
        lo, hi, ow = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().gorse_topk_tri_slice(self.h, rank, C.byref(lo), C.byref(hi), C.byref(ow)))
        return lo.value, hi.value, ow.value

    def tri_thresholds_get(self, lo, hi):
        out = np.empty(hi - lo, np.float32)
        check(lib().gorse_topk_tri_thresholds_get(self.h, lo, hi, out.ctypes.data_as(_vp)))
        return out

    def tri_thresholds_put(self, lo, hi, thr):
        thr = _arr(thr, np.float32)
        assert thr.size == hi - lo
        check(lib().gorse_topk_tri_thresholds_put(self.h, lo, hi, thr.ctypes.data_as(_vp)))

    def tri_sweep(self):
        check(lib().gorse_topk_tri_sweep(self.h))

    def tri_pack_device(self, dest):
        """builds the message for `dest` in the handle's device buffers; returns (number of counts, number of entries)"""
        nc, ne = C.c_int64(0), C.c_int64(0)
        check(lib().gorse_topk_tri_pack(self.h, dest, C.byref(nc), C.byref(ne)))
        return nc.value, ne.value

    def tri_pack_fetch(self, nc, ne):
        counts, entries = np.empty(nc, np.int32), np.empty(ne, np.uint64)
        check(lib().gorse_topk_tri_pack_read(self.h, counts.ctypes.data_as(_vp), entries.ctypes.data_as(_vp)))
        return counts, entries

    def tri_pack(self, dest):
        """the message for `dest`: (counts int32 per query dest owns, entries uint64 = (key, row) pairs end to end)"""
        return self.tri_pack_fetch(*self.tri_pack_device(dest))

    def tri_unpack(self, src, counts, entries):
        counts, entries = _arr(counts, np.int32), _arr(entries, np.uint64)
        check(lib().gorse_topk_tri_unpack(self.h, src, counts.ctypes.data_as(_vp), counts.size, entries.ctypes.data_as(_vp), entries.size))

    # device-pointer forms (the messages stay on the GPU: torch tensors' data_ptr(); gorse_amd.dist.HipTriEngine(device="cuda"))
    def tri_thresholds_get_ptr(self, lo, hi, ptr):
        check(lib().gorse_topk_tri_thresholds_get(self.h, lo, hi, _vp(ptr)))

    def tri_thresholds_put_ptr(self, lo, hi, ptr):
        check(lib().gorse_topk_tri_thresholds_put(self.h, lo, hi, _vp(ptr)))

    def tri_pack_read_ptr(self, counts_ptr, entries_ptr):
        check(lib().gorse_topk_tri_pack_read(self.h, _vp(counts_ptr), _vp(entries_ptr)))

    def tri_unpack_ptr(self, src, counts_ptr, n_counts, entries_ptr, n_entries):
        check(lib().gorse_topk_tri_unpack(self.h, src, _vp(counts_ptr), n_counts, _vp(entries_ptr), n_entries))

    def tri_finish(self, idx=None, dist=None, fetch=True):
        """rescoring + tie path of the queries this rank owns; with fetch their rows are written into idx / dist (nq x k, allocated
        here when not given: the rows of other ranks' queries are then -1 / +inf)"""
        k, nq = self._tri
        if fetch and idx is None:
            idx, dist = np.full((nq, k), -1, np.int32), np.full((nq, k), np.inf, np.float32)
        check(lib().gorse_topk_tri_finish(self.h, _p(idx, _i32p) if fetch else None, _p(dist, _f32p) if fetch else None))
        return idx, dist

    def set_mask(self, admissible=None):
        m = None if admissible is None else _arr(admissible, np.uint8)
        if m is not None and m.size != self.N:
            raise GorseHipError(ERR_INVALID, "mask must have N entries")
        check(lib().gorse_topk_set_mask(self.h, None if m is None else m.ctypes.data_as(_vp)))

    def resweeps(self):
        n = C.c_int64(0)
        check(lib().gorse_hip_test_topk_resweeps(self.h, C.byref(n)))
        return n.value

    def pilot_state(self, n):
        f, c = np.empty(n, np.uint8), np.empty(n, np.int32)
        check(lib().gorse_hip_test_topk_get_pilot_state(self.h, f.ctypes.data_as(C.POINTER(C.c_uint8)), c.ctypes.data_as(_i32p), n))
        return f, c

    def warm_thresholds(self, n):
        out = np.empty(n, np.float32)
        check(lib().gorse_hip_test_topk_get_thresholds(self.h, out.ctypes.data_as(_f32p), n))
        return out

    def last_flags(self, n):
        """per-query flags of the last MFMA search's last chunk behind the rescoring (non-zero: the query took the tie path)"""
        f = np.empty(n, np.uint8)
        check(lib().gorse_hip_test_topk_get_flags(self.h, f.ctypes.data_as(C.POINTER(C.c_uint8)), n))
        return f

    def foreign_counts(self, n):
        """entries appended to each query's foreign list by the last symmetric sweep (> 512: the list overflowed)"""
        c = np.empty(n, np.int32)
        check(lib().gorse_hip_test_topk_get_foreign_counts(self.h, c.ctypes.data_as(_i32p), n))
        return c

    def sym_stats(self):
        out = (C.c_uint64 * 4)()
        check(lib().gorse_hip_test_topk_sym_stats(self.h, out))
        return [int(x) for x in out]

    def last_symmetric(self):
        s = C.c_int32(0)
        check(lib().gorse_hip_test_topk_last_symmetric(self.h, C.byref(s)))
        return bool(s.value)

    def synchronize(self):
        check(lib().gorse_topk_synchronize(self.h))

    def set_profiling(self, on):
        check(lib().gorse_topk_set_profiling(self.h, int(bool(on))))

    def get_profile(self, cls):
        n, ms = C.c_int64(0), C.c_double(0)
        check(lib().gorse_topk_get_profile(self.h, cls, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def sweep_profile(self):
        out = (C.c_uint64 * 16)()
        check(lib().gorse_hip_test_get_sweep_profile(self.h, out))
        return [int(x) for x in out]

    def last_stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        check(lib().gorse_topk_last_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value


def topk_tri_all_pairs_local(handles, k, q_begin=0, q_end=None, fetch=True):
    """gorse_topk_tri_all_pairs_local: the triangle-sharded all-pairs search over the TopK handles of ONE process (handles[r] = rank r)"""
    q_end = handles[0].N if q_end is None else q_end
    nq = q_end - q_begin
    hs = (_vp * len(handles))(*[h.h for h in handles])
    idx = np.empty((nq, k), np.int32) if fetch else None
    dist = np.empty((nq, k), np.float32) if fetch else None
    check(lib().gorse_topk_tri_all_pairs_local(hs, len(handles), q_begin, q_end, k, _p(idx, _i32p), _p(dist, _f32p)))
    return idx, dist


class Comm:
    """One rank of an RCCL communicator owned by the library (gorse_comm_*): Comm.unique_id() on rank 0, the bytes shipped
    to every rank by the caller, Comm(id, world, rank, device) everywhere; Comm.local(devices) = all ranks in this process."""

    def __init__(self, uid=None, world=1, rank=0, device=0, _handle=None):
        if _handle is not None:
            self.h = _handle
        else:
            buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(uid))
            self.h = _vp()
            check(lib().gorse_comm_create(C.byref(self.h), C.cast(buf, _vp), world, rank, device))
        w, r = C.c_int32(0), C.c_int32(0)
        check(lib().gorse_comm_info(self.h, C.byref(w), C.byref(r)))
        self.world, self.rank = w.value, r.value

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        check(lib().gorse_comm_unique_id(C.cast(buf, _vp)))
        return bytes(buf)

    @staticmethod
    def local(devices):
        devs = _arr(devices, np.int32)
        out = (_vp * devs.size)()
        check(lib().gorse_comm_create_local(out, _p(devs, _i32p), devs.size))
        return [Comm(_handle=_vp(out[i])) for i in range(devs.size)]

    def close(self):
        if getattr(self, "h", None):
            lib().gorse_comm_destroy(self.h)
            self.h = None

    __del__ = close

    def allreduce_f32(self, values):
        a = np.array(values, dtype=np.float32)
        check(lib().gorse_comm_allreduce_f32(self.h, _p(a, _f32p), a.size))
        return a


def comm_available():
    """None when RCCL can be opened in this process, else the reason"""
    if lib().gorse_comm_available() == 0:
        return None
    return lib().gorse_hip_last_error().decode()


def allreduce_f32_local(comms, arrays):
    """gorse_comm_allreduce_f32_local: one array of equal length per communicator of this process, summed in place"""
    arrs = [np.ascontiguousarray(a, np.float32) for a in arrays]
    cs = (_vp * len(comms))(*[c.h for c in comms])
    bs = (_f32p * len(arrs))(*[_p(a, _f32p) for a in arrs])
    check(lib().gorse_comm_allreduce_f32_local(cs, len(comms), bs, arrs[0].size))
    return arrs


def _pairs(mfs, comms):
    n = len(mfs)
    hs, cs = (_vp * n)(*[m.h for m in mfs]), (_vp * n)(*[c.h for c in comms])
    return hs, cs, n


def item_allreduce(mfs, comms):
    """gorse_mf_item_allreduce over this process's (handle, communicator) pairs"""
    hs, cs, n = _pairs(mfs, comms)
    check(lib().gorse_mf_item_allreduce(hs, cs, n))


def rows_allgather(mfs, comms, side, row_splits):
    hs, cs, n = _pairs(mfs, comms)
    sp = _arr(row_splits, np.int64)
    check(lib().gorse_mf_rows_allgather(hs, cs, n, side, _p(sp, _i64p)))


class Sparse:
    """One gorse_sparse handle: N sparse vectors (CSR, strictly ascending indices per row) resident on one GPU."""

    def __init__(self, indptr, indices, values, device=0):
        self.indptr = _arr(indptr, np.int64)
        self.indices = _arr(indices, np.uint32)
        self.values = _arr(values, np.float32)
        self.N = self.indptr.size - 1
        self.h = _vp()
        check(lib().gorse_sparse_create(C.byref(self.h), device, self.N, _p(self.indptr, _i64p),
                                        self.indices.ctypes.data_as(_vp), _p(self.values, _f32p)))

    def close(self):
        if getattr(self, "h", None):
            lib().gorse_sparse_destroy(self.h)
            self.h = None

    __del__ = close

    # ---- the triangle-sharded all-pairs search (include/gorse_hip.h, gorse_topk_tri_*): host-memory forms of the messages ----
    def tri_begin(self, k, rank, world, q_begin=0, q_end=None):
        q_end = self.N if q_end is None else q_end
        self._tri = (k, q_end - q_begin)
        check(lib().gorse_topk_tri_begin(self.h, q_begin, q_end, k, rank, world))

    def tri_slice(self, rank):
        """(lo, hi, owned): the queries whose pilots `rank` runs, and how many queries it owns"""
        lo, hi, ow = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(lib().gorse_topk_tri_slice(self.h, rank, C.byref(lo), C.byref(hi), C.byref(ow)))
        return lo.value, hi.value, ow.value

    def tri_thresholds_get(self, lo, hi):
        out = np.empty(hi - lo, np.float32)
        check(lib().gorse_topk_tri_thresholds_get(self.h, lo, hi, out.ctypes.data_as(_vp)))
        return out

    def tri_thresholds_put(self, lo, hi, thr):
        thr = _arr(thr, np.float32)
        assert thr.size == hi - lo
        check(lib().gorse_topk_tri_thresholds_put(self.h, lo, hi, thr.ctypes.data_as(_vp)))

    def tri_sweep(self):
        check(lib().gorse_topk_tri_sweep(self.h))

    def tri_pack_device(self, dest):
        """builds the message for `dest` in the handle's device buffers; returns (number of counts, number of entries)"""
        nc, ne = C.c_int64(0), C.c_int64(0)
        check(lib().gorse_topk_tri_pack(self.h, dest, C.byref(nc), C.byref(ne)))
        return nc.value, ne.value

    def tri_pack_fetch(self, nc, ne):
        counts, entries = np.empty(nc, np.int32), np.empty(ne, np.uint64)
        check(lib().gorse_topk_tri_pack_read(self.h, counts.ctypes.data_as(_vp), entries.ctypes.data_as(_vp)))
        return counts, entries

    def tri_pack(self, dest):
        """the message for `dest`: (counts int32 per query dest owns, entries uint64 = (key, row) pairs end to end)"""
        return self.tri_pack_fetch(*self.tri_pack_device(dest))

    def tri_unpack(self, src, counts, entries):
        counts, entries = _arr(counts, np.int32), _arr(entries, np.uint64)
        check(lib().gorse_topk_tri_unpack(self.h, src, counts.ctypes.data_as(_vp), counts.size, entries.ctypes.data_as(_vp), entries.size))

    # device-pointer forms (the messages stay on the GPU: torch tensors' data_ptr(); gorse_amd.dist.HipTriEngine(device="cuda"))
    def tri_thresholds_get_ptr(self, lo, hi, ptr):
        check(lib().gorse_topk_tri_thresholds_get(self.h, lo, hi, _vp(ptr)))

    def tri_thresholds_put_ptr(self, lo, hi, ptr):
        check(lib().gorse_topk_tri_thresholds_put(self.h, lo, hi, _vp(ptr)))

    def tri_pack_read_ptr(self, counts_ptr, entries_ptr):
        check(lib().gorse_topk_tri_pack_read(self.h, _vp(counts_ptr), _vp(entries_ptr)))

    def tri_unpack_ptr(self, src, counts_ptr, n_counts, entries_ptr, n_entries):
        check(lib().gorse_topk_tri_unpack(self.h, src, _vp(counts_ptr), n_counts, _vp(entries_ptr), n_entries))

    def tri_finish(self, idx=None, dist=None, fetch=True):
        """rescoring + tie path of the queries this rank owns; with fetch their rows are written into idx / dist (nq x k, allocated
        here when not given: the rows of other ranks' queries are then -1 / +inf)"""
        k, nq = self._tri
        if fetch and idx is None:
            idx, dist = np.full((nq, k), -1, np.int32), np.full((nq, k), np.inf, np.float32)
        check(lib().gorse_topk_tri_finish(self.h, _p(idx, _i32p) if fetch else None, _p(dist, _f32p) if fetch else None))
        return idx, dist

    def set_mask(self, admissible=None):
        m = None if admissible is None else _arr(admissible, np.uint8)
        if m is not None and m.size != self.N:
            raise GorseHipError(ERR_INVALID, "mask must have N entries")
        check(lib().gorse_sparse_set_mask(self.h, None if m is None else m.ctypes.data_as(_vp)))

    def search(self, q_indptr, q_indices, q_values, k, exclude=None):
        qp = _arr(q_indptr, np.int64)
        qi = _arr(q_indices, np.uint32)
        qv = _arr(q_values, np.float32)
        nq = qp.size - 1
        ex = None if exclude is None else _arr(exclude, np.int64)
        idx, sc, cnt = np.empty((nq, k), np.int32), np.empty((nq, k), np.float32), np.empty(nq, np.int32)
        check(lib().gorse_sparse_search(self.h, nq, _p(qp, _i64p), qi.ctypes.data_as(_vp), _p(qv, _f32p), _p(ex, _i64p), k,
                                        _p(idx, _i32p), _p(sc, _f32p), _p(cnt, _i32p)))
        return idx, sc, cnt

    def all_pairs(self, k, q_begin=0, q_end=None, exclude_self=True, fetch=True):
        q_end = self.N if q_end is None else q_end
        nq = q_end - q_begin
        idx = np.empty((nq, k), np.int32) if fetch else None
        sc = np.empty((nq, k), np.float32) if fetch else None
        cnt = np.empty(nq, np.int32) if fetch else None
        check(lib().gorse_sparse_all_pairs(self.h, q_begin, q_end, k, int(bool(exclude_self)), _p(idx, _i32p),
                                           _p(sc, _f32p), _p(cnt, _i32p)))
        return idx, sc, cnt

    def synchronize(self):
        check(lib().gorse_sparse_synchronize(self.h))

    def set_profiling(self, on):
        check(lib().gorse_sparse_set_profiling(self.h, int(bool(on))))

    def get_profile(self):
        n, ms = C.c_int64(0), C.c_double(0)
        check(lib().gorse_sparse_get_profile(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def last_stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        check(lib().gorse_sparse_last_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sym_stats(self):
        """the last call: (ran in symmetric form, rows redone after a foreign list overflowed, foreign entries ranked, longest foreign list)"""
        out = (C.c_int64 * 4)()
        lib().gorse_hip_test_sparse_sym_stats(self.h, out)
        return tuple(int(x) for x in out)

    def trace(self, on=True):
        """probe: switch the per-work-item records on / off; returns the records of the last call (n x 16 uint64)"""
        n = lib().gorse_hip_test_sparse_trace(self.h, int(bool(on)), None, 0)
        out = np.zeros((max(n, 0), 16), np.uint64)
        if n > 0:
            lib().gorse_hip_test_sparse_trace(self.h, int(bool(on)), out.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        return out


def sgemm(transA, transB, m, n, k, a, lda, b, ldb, c, ldc, device=0):
    a, b = _arr(a, np.float32), _arr(b, np.float32)
    c = np.array(c, dtype=np.float32, order="C")
    check(lib().gorse_hip_sgemm(device, int(transA), int(transB), m, n, k, _p(a, _f32p), lda, _p(b, _f32p), ldb,
                                _p(c, _f32p), ldc))
    return c


def sgemm_device(transA, transB, m, n, k, a_ptr, lda, b_ptr, ldb, c_ptr, ldc, device=0):
    """gorse_hip_sgemm_device: the operands are device addresses (e.g. torch tensors' data_ptr()); C is updated in place"""
    check(lib().gorse_hip_sgemm_device(device, int(transA), int(transB), m, n, k, C.c_void_p(a_ptr), lda, C.c_void_p(b_ptr), ldb,
                                       C.c_void_p(c_ptr), ldc))
