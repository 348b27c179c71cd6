"""Python face of the C++ host mirror (gorse_amd/host, libgorse_host.so): the reference's
model/cf, dataset, common/heap and common/ann interfaces with their Go names, so that the parity
tests read like model/cf/model_test.go, evaluator_test.go, heap/*_test.go and ann_test.go.
All numerics run on the MI355X through the C ABI; nothing here computes on the CPU.
"""
import ctypes as C
import os

import numpy as np

from . import capi

HOST_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgorse_host.so")
_f32p, _i32p = C.POINTER(C.c_float), C.POINTER(C.c_int32)
_H = None
NDCG, Precision, Recall, HR, MAP, MRR = range(6)


def host():
    global _H
    if _H is None:
        capi.lib()  # libgorse_hip.so first (RTLD_GLOBAL), then the host library that links against it
        if not os.path.exists(HOST_LIB):
            raise RuntimeError("%s not found: run __graft_entry__.build()" % HOST_LIB)
        H = C.CDLL(HOST_LIB)
        H.gh_last_error.restype = C.c_char_p
        for n in ("gh_dataset_new", "gh_bpr_new", "gh_als_new", "gh_model_unmarshal", "gh_bruteforce_new",
                  "gh_dataset_new_shared"):
            getattr(H, n).restype = C.c_void_p
        H.gh_model_name.restype = C.c_char_p
        H.gh_model_marshal.restype = C.c_int64
        H.gh_metric.restype = C.c_float
        _H = H
    return _H


class HostError(RuntimeError):
    def __init__(self, code):
        super().__init__("host error %d: %s" % (code, host().gh_last_error().decode("utf-8", "replace")))
        self.code = code


def _ck(rc):
    if rc != 0:
        raise HostError(rc)


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


class Dataset:
    """dataset.Dataset as a CFSplit (dataset/dataset.go:40-59)."""

    def __init__(self, share_dicts_with=None):
        H = host()
        self.p = C.c_void_p(H.gh_dataset_new_shared(share_dicts_with.p) if share_dicts_with is not None
                            else H.gh_dataset_new())

    def __del__(self):
        if getattr(self, "p", None):
            host().gh_dataset_free(self.p)
            self.p = None

    def AddUser(self, user_id):
        host().gh_dataset_add_user(self.p, str(user_id).encode())

    def AddItem(self, item_id):
        host().gh_dataset_add_item(self.p, str(item_id).encode())

    def AddFeedback(self, user_id, item_id):
        host().gh_dataset_add_feedback_str(self.p, str(user_id).encode(), str(item_id).encode())

    def add_feedback_arrays(self, u, i):
        u, i = _i32(u), _i32(i)
        host().gh_dataset_add_feedback(self.p, u.ctypes.data_as(_i32p), i.ctypes.data_as(_i32p), C.c_int64(u.size))

    def SetNegatives(self, user, negs):
        negs = _i32(negs)
        host().gh_dataset_set_negatives(self.p, int(user), negs.ctypes.data_as(_i32p), negs.size)

    def CountUsers(self):
        return host().gh_dataset_count_users(self.p)

    def CountItems(self):
        return host().gh_dataset_count_items(self.p)

    def CountFeedback(self):
        return host().gh_dataset_count_feedback(self.p)

    @classmethod
    def _wrap(cls, ptr):
        d = cls.__new__(cls)
        d.p = C.c_void_p(ptr)
        return d

    def SplitCF(self, numTestUsers, seed):
        """dataset.go:258-318: (train, test) by user-leave-one-out; the production split is SplitCF(0, 0) (master/tasks.go:232)"""
        a, b = C.c_void_p(), C.c_void_p()
        H = host()
        H.gh_dataset_split_cf.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        _ck(H.gh_dataset_split_cf(self.p, int(numTestUsers), int(seed), C.byref(a), C.byref(b)))
        return Dataset._wrap(a.value), Dataset._wrap(b.value)

    @staticmethod
    def LoadNCF(train_text, test_text):
        """LoadDataFromBuiltIn (dataset.go:398-490) on the contents of train.txt / test.txt"""
        a, b = C.c_void_p(), C.c_void_p()
        H = host()
        H.gh_dataset_load_ncf.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        _ck(H.gh_dataset_load_ncf(train_text.encode(), test_text.encode(), C.byref(a), C.byref(b)))
        return Dataset._wrap(a.value), Dataset._wrap(b.value)

    def _row(self, side, row):
        H = host()
        H.gh_dataset_row.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _i32p, C.c_int32]
        n = H.gh_dataset_row(self.p, side, int(row), None, 0)
        out = np.zeros(max(n, 1), np.int32)
        H.gh_dataset_row(self.p, side, int(row), out.ctypes.data_as(_i32p), out.size)
        return out[:n].tolist()

    def GetUserFeedback(self):
        return [self._row(0, u) for u in range(self.CountUsers())]

    def GetItemFeedback(self):
        return [self._row(1, i) for i in range(self.CountItems())]

    def Negatives(self, user):
        return self._row(2, user)

    def _idf(self, side, n):
        H = host()
        H.gh_dataset_idf.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.c_int32]
        out = np.zeros(max(n, 1), np.float32)
        m = H.gh_dataset_idf(self.p, side, out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
        return out[:m].copy()

    def GetUserIDF(self):
        """dataset/dataset.go:160-166: log(1 + #items / freq(user)), the weights of the "users" item-to-item vectors"""
        return self._idf(0, self.CountUsers())

    def GetItemIDF(self):
        """dataset/dataset.go:174-180: log(1 + #users / freq(item)), the weights of the "items" user-to-user vectors"""
        return self._idf(1, self.CountItems())


def datasets_from_synth(data, preload_negatives=True):
    """(train, test) Datasets from a gorse_amd.synth.CFData, in the NCF built-in layout
    (dataset.LoadDataFromBuiltIn, dataset/dataset.go:398-490): shared dictionaries, preloaded negatives -- or, with
    preload_negatives=False, a production split (master/tasks.go:232): Evaluate samples the negatives itself."""
    train = Dataset()
    for u in range(data.U):
        train.AddUser(u)
    for i in range(data.I):
        train.AddItem(i)
    rows = np.repeat(np.arange(data.U, dtype=np.int32), np.diff(data.uptr))
    train.add_feedback_arrays(rows, data.uidx)
    test = Dataset(share_dicts_with=train)
    trows = np.repeat(np.arange(data.U, dtype=np.int32), np.diff(data.test_ptr))
    test.add_feedback_arrays(trows, data.test_idx)
    if preload_negatives:
        for u in np.nonzero(np.diff(data.neg_ptr) > 0)[0]:
            test.SetNegatives(int(u), data.neg_idx[data.neg_ptr[u]:data.neg_ptr[u + 1]])
    return train, test


class FitConfig:
    """cf.FitConfig (model/cf/model.go:50-80)."""

    def __init__(self):
        self.Jobs, self.Verbose, self.Candidates, self.TopK, self.Patience = 1, 10, 100, 10, 0
        self.cancel = None

    def SetVerbose(self, v):
        self.Verbose = v
        return self

    def SetJobs(self, j):
        self.Jobs = j
        return self

    def SetPatience(self, p):
        self.Patience = p
        return self


def NewFitConfig():
    return FitConfig()


class Score:
    def __init__(self, ndcg, precision, recall):
        self.NDCG, self.Precision, self.Recall = ndcg, precision, recall


class _Model:
    def __init__(self, ctor, params, ptr=None):
        H = host()
        if ptr is not None:
            self.p = C.c_void_p(ptr)
        else:
            names = (C.c_char_p * len(params))(*[k.encode() for k in params])
            vals = (C.c_double * len(params))(*[float(v) for v in params.values()])
            self.p = C.c_void_p(getattr(H, ctor)(names, vals, len(params)))
        self.log = ""
        self.epochs_done = 0

    def __del__(self):
        if getattr(self, "p", None):
            host().gh_model_free(self.p)
            self.p = None

    def Fit(self, trainSet, valSet, config):
        score = (C.c_float * 3)()
        done = C.c_int32(0)
        logbuf = C.create_string_buffer(1 << 20)
        cancel = config.cancel.ctypes.data_as(_i32p) if config.cancel is not None else None
        _ck(host().gh_model_fit(self.p, trainSet.p, valSet.p, config.Jobs, config.Verbose, config.Candidates, config.TopK,
                                config.Patience, cancel, score, C.byref(done), logbuf, C.c_int64(len(logbuf))))
        self.log = logbuf.value.decode()
        self.epochs_done = done.value
        return Score(score[0], score[1], score[2])

    def Predict(self, user_id, item_id):
        out = C.c_float(0)
        _ck(host().gh_model_predict(self.p, str(user_id).encode(), str(item_id).encode(), C.byref(out)))
        return out.value

    def internalPredict(self, u, i):
        out = C.c_float(0)
        _ck(host().gh_model_internal_predict(self.p, int(u), int(i), C.byref(out)))
        return out.value

    def GetUserFactor(self, u):
        out = np.empty(host().gh_model_n_factors(self.p), np.float32)
        host().gh_model_get_user_factor(self.p, int(u), out.ctypes.data_as(_f32p))
        return out

    def GetItemFactor(self, i):
        out = np.empty(host().gh_model_n_factors(self.p), np.float32)
        host().gh_model_get_item_factor(self.p, int(i), out.ctypes.data_as(_f32p))
        return out

    def IsUserPredictable(self, u):
        return bool(host().gh_model_is_user_predictable(self.p, int(u)))

    def IsItemPredictable(self, i):
        return bool(host().gh_model_is_item_predictable(self.p, int(i)))

    def CountUsers(self):
        return host().gh_model_count_users(self.p)

    def CountItems(self):
        return host().gh_model_count_items(self.p)

    def UserIndex(self, user_id):
        return host().gh_model_user_index(self.p, str(user_id).encode())

    def ItemIndex(self, item_id):
        return host().gh_model_item_index(self.p, str(item_id).encode())

    def Clear(self):
        host().gh_model_clear(self.p)

    def Invalid(self):
        return bool(host().gh_model_invalid(self.p))

    def Name(self):
        return host().gh_model_name(self.p).decode()

    def factors(self):
        """(P, Q): the model's UserFactor / ItemFactor rows as matrices"""
        U, I = self.CountUsers(), self.CountItems()
        return (np.stack([self.GetUserFactor(u) for u in range(U)]), np.stack([self.GetItemFactor(i) for i in range(I)]))

    def load_factors(self, P, Q):
        P, Q = np.ascontiguousarray(P, np.float32), np.ascontiguousarray(Q, np.float32)
        _ck(host().gh_model_load_factors(self.p, P.shape[0], Q.shape[0], P.ctypes.data_as(_f32p), Q.ctypes.data_as(_f32p)))


class BPR(_Model):
    def __init__(self, params=None, ptr=None):
        super().__init__("gh_bpr_new", params or {}, ptr)


class ALS(_Model):
    def __init__(self, params=None, ptr=None):
        super().__init__("gh_als_new", params or {}, ptr)


def NewBPR(params):
    return BPR(params)


def NewALS(params):
    return ALS(params)


def _search_out():
    return C.create_string_buffer(64), C.create_string_buffer(4096), (C.c_float * 3)()


def _search_result(type_buf, params_buf, score):
    params = {}
    for line in params_buf.value.decode().splitlines():
        k, v = line.split("=")
        params[k] = float(v)
    return {"Type": type_buf.value.decode(), "Params": params, "Score": Score(score[0], score[1], score[2])}


def search_mock(n_trials, seed=0, sampler="tpe"):
    """optimize_test.go's TestTPE search (mock model, NDCG = NFactors + InitMean + InitStdDev) under the TPE study of the
    reference's test (or independent random trials); returns (study.GetBestValue(), search.Result())"""
    t, p, sc = _search_out()
    best = C.c_double(0)
    H = host()
    H.gh_search_mock.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double), C.c_char_p, C.c_int64, C.c_char_p, C.c_int64,
                                 C.POINTER(C.c_float)]
    _ck(H.gh_search_mock(n_trials, seed, 0 if sampler == "tpe" else 1, C.byref(best), t, len(t), p, len(p), sc))
    return best.value, _search_result(t, p, sc)


def ModelSearch(trainSet, valSet, n_trials, seed=0, jobs=1, patience=0, overrides=None, keep_resident=True, cancel=None):
    """optimizeCollaborativeFiltering (master/tasks.go:1268-1316): BPR and ALS at their defaults (+ overrides such as
    {"NEpochs": 20}) over n_trials random trials; the training set stays on the device across the trials when
    keep_resident.  Returns (search.Result(), {"uploads", "reuses", "trials"})."""
    overrides = overrides or {}
    names = (C.c_char_p * len(overrides))(*[k.encode() for k in overrides])
    vals = (C.c_double * len(overrides))(*[float(v) for v in overrides.values()])
    t, p, sc = _search_out()
    counters = (C.c_int32 * 3)()
    H = host()
    H.gh_model_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_char_p),
                                  C.POINTER(C.c_double), C.c_int32, C.c_int32, _i32p, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64,
                                  C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    cptr = cancel.ctypes.data_as(_i32p) if cancel is not None else None
    _ck(H.gh_model_search(trainSet.p, valSet.p, n_trials, seed, jobs, patience, names, vals, len(overrides), int(keep_resident),
                          cptr, t, len(t), p, len(p), sc, counters))
    return _search_result(t, p, sc), {"uploads": counters[0], "reuses": counters[1], "trials": counters[2]}


def MarshalModel(m):
    n = host().gh_model_marshal(m.p, None, C.c_int64(0))
    buf = C.create_string_buffer(n)
    host().gh_model_marshal(m.p, buf, C.c_int64(n))
    return buf.raw


def UnmarshalModel(data):
    ptr = host().gh_model_unmarshal(data, C.c_int64(len(data)))
    if not ptr:
        raise HostError(-100)
    name = host().gh_model_name(C.c_void_p(ptr)).decode()
    return (BPR if name == "bpr" else ALS)(ptr=ptr)


def Evaluate(estimator, testSet, trainSet, topK, numCandidates, nJobs, *scorers):
    ids = _i32(list(scorers))
    out = np.zeros(ids.size, np.float32)
    _ck(host().gh_evaluate(estimator.p, testSet.p, trainSet.p, topK, numCandidates, nJobs, ids.ctypes.data_as(_i32p),
                           ids.size, out.ctypes.data_as(_f32p)))
    return out


def metric(mid, target, rank):
    t, r = _i32(list(target)), _i32(list(rank))
    return float(host().gh_metric(mid, t.ctypes.data_as(_i32p), t.size, r.ctypes.data_as(_i32p), r.size))


class TopKFilter:
    """heap.TopKFilter[int32, float32] (common/heap/filter.go)."""

    def __init__(self, k):
        self.k, self.items, self.weights = k, [], []

    def Push(self, item, weight):
        self.items.append(item)
        self.weights.append(weight)

    def PopAll(self):
        it, w = _i32(self.items), np.ascontiguousarray(self.weights, np.float32)
        oi, ow = np.zeros(self.k + 1, np.int32), np.zeros(self.k + 1, np.float32)
        n = host().gh_topk_filter(self.k, it.ctypes.data_as(_i32p), w.ctypes.data_as(_f32p), it.size,
                                  oi.ctypes.data_as(_i32p), ow.ctypes.data_as(_f32p))
        return [(int(a), float(b)) for a, b in zip(oi[:n], ow[:n])]

    def PopAllValues(self):
        return [v for v, _ in self.PopAll()]


def priority_queue_drain(desc, elements, weights, reverse=False):
    """heap.PriorityQueue: push all, (optionally Reverse()), pop all (common/heap/pq.go)."""
    it, w = _i32(elements), np.ascontiguousarray(weights, np.float32)
    oi, ow = np.zeros(it.size + 1, np.int32), np.zeros(it.size + 1, np.float32)
    n = host().gh_pq_drain(int(desc), int(reverse), it.ctypes.data_as(_i32p), w.ctypes.data_as(_f32p), it.size,
                           oi.ctypes.data_as(_i32p), ow.ctypes.data_as(_f32p))
    if n < 0:
        raise HostError(n)
    return [(int(a), float(b)) for a, b in zip(oi[:n], ow[:n])]


class Bruteforce:
    """ann.Bruteforce (common/ann/bruteforce.go) with the scan on the GPU; an ann.Index."""

    def __init__(self, metric):
        self.p = C.c_void_p(host().gh_bruteforce_new(metric))

    def __del__(self):
        if getattr(self, "p", None):
            host().gh_bruteforce_free(self.p)
            self.p = None

    def Add(self, v):
        v = np.ascontiguousarray(v, np.float32)
        ret = C.c_int32(0)
        _ck(host().gh_bruteforce_add(self.p, v.ctypes.data_as(_f32p), v.size, C.byref(ret)))
        return ret.value

    def SearchIndex(self, q, k, prune0):
        idx, dist, cnt = np.zeros(k, np.int32), np.zeros(k, np.float32), C.c_int32(0)
        _ck(host().gh_bruteforce_search_index(self.p, int(q), k, int(prune0), idx.ctypes.data_as(_i32p),
                                              dist.ctypes.data_as(_f32p), C.byref(cnt)))
        return [(int(a), float(b)) for a, b in zip(idx[:cnt.value], dist[:cnt.value])]

    def SearchVector(self, q, k, prune0):
        q = np.ascontiguousarray(q, np.float32)
        idx, dist, cnt = np.zeros(k, np.int32), np.zeros(k, np.float32), C.c_int32(0)
        _ck(host().gh_bruteforce_search_vector(self.p, q.ctypes.data_as(_f32p), q.size, k, int(prune0),
                                               idx.ctypes.data_as(_i32p), dist.ctypes.data_as(_f32p), C.byref(cnt)))
        return [(int(a), float(b)) for a, b in zip(idx[:cnt.value], dist[:cnt.value])]
