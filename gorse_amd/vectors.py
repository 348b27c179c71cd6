"""Python face of the C++ twin of vectors.Database (gorse_amd/host/gorse_vectors.hpp; storage/vectors/database.go:107-120)
so that the parity tests read like storage/vectors/database_test.go.  `Open("hip://")` searches on the MI355X through
libgorse_hip; `Database(searcher=..., sparse_searcher=...)` takes search callbacks instead (the CPU test-suite injects
checkers built on the oracle there -- the product path never does)."""
import ctypes as C
import datetime as dt

import numpy as np

from . import cf

Cosine, Euclidean, Dot = 0, 1, 2  # database.go:29-33


class ErrNotFound(RuntimeError):  # storage/errors.go:20
    pass


class ErrAlreadyExists(RuntimeError):  # storage/errors.go:24
    pass


class ErrNotSupported(RuntimeError):  # storage/errors.go:23
    pass


_ERR = {-201: ErrNotFound, -202: ErrAlreadyExists, -203: ErrNotSupported}
_INVALID = -1  # GORSE_ERR_INVALID (std::invalid_argument in the C++ twin)
SEARCH_CB = C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_float), C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int64,
                        C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int32))
SPARSE_CB = C.CFUNCTYPE(C.c_int32, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                        C.POINTER(C.c_uint8), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.POINTER(C.c_float),
                        C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_int32))
_H = None


def _host():
    global _H
    if _H is None:
        H = cf.host()
        H.gh_vdb_open.restype = C.c_void_p
        H.gh_vdb_open.argtypes = [C.c_char_p]
        H.gh_vdb_open_with_searcher.restype = C.c_void_p
        H.gh_vdb_open_with_searcher.argtypes = [SEARCH_CB]
        H.gh_vdb_open_with_searchers.restype = C.c_void_p
        H.gh_vdb_open_with_searchers.argtypes = [C.c_void_p, C.c_void_p]
        H.gh_vdb_free.argtypes = [C.c_void_p]
        for n, args in (("gh_vdb_close", [C.c_void_p]),
                        ("gh_vdb_add_collection", [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]),
                        ("gh_vdb_delete_collection", [C.c_void_p, C.c_char_p]),
                        ("gh_vdb_describe", [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
                        ("gh_vdb_add_staged", [C.c_void_p, C.c_char_p]),
                        ("gh_vdb_get", [C.c_void_p, C.c_char_p, C.c_char_p]),
                        ("gh_vdb_delete_vectors", [C.c_void_p, C.c_char_p, C.c_int64]),
                        ("gh_vdb_query_staged", [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32]),
                        ("gh_vdb_query_sparse_staged", [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32]),
                        ("gh_vdb_query_batch", [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int64, C.c_int32, C.c_char_p, C.c_int32])):
            getattr(H, n).restype = C.c_int32
            getattr(H, n).argtypes = args
        H.gh_vdb_list.restype = C.c_int64
        H.gh_vdb_list.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        H.gh_vdb_count.restype = C.c_int64
        H.gh_vdb_count.argtypes = [C.c_void_p, C.c_char_p]
        H.gh_vdb_stage_vector.restype = None
        H.gh_vdb_stage_vector.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_uint32), C.c_int32,
                                          C.c_int32, C.c_char_p, C.c_int64]
        H.gh_vdb_stage_clear.restype = None
        H.gh_vdb_result_count.restype = C.c_int64
        H.gh_vdb_result_split.restype = C.c_int64
        H.gh_vdb_result_split.argtypes = [C.c_int64]
        H.gh_vdb_result_id.restype = C.c_char_p
        H.gh_vdb_result_id.argtypes = [C.c_int64]
        H.gh_vdb_result_score.restype = C.c_float
        H.gh_vdb_result_score.argtypes = [C.c_int64]
        H.gh_vdb_result_hidden.argtypes = [C.c_int64]
        H.gh_vdb_result_timestamp.restype = C.c_int64
        H.gh_vdb_result_timestamp.argtypes = [C.c_int64]
        H.gh_vdb_result_dim.argtypes = [C.c_int64]
        H.gh_vdb_result_values.restype = None
        H.gh_vdb_result_values.argtypes = [C.c_int64, C.POINTER(C.c_float)]
        H.gh_vdb_result_nnz.argtypes = [C.c_int64]
        H.gh_vdb_result_indices.restype = None
        H.gh_vdb_result_indices.argtypes = [C.c_int64, C.POINTER(C.c_uint32)]
        H.gh_vdb_result_categories.restype = C.c_int64
        H.gh_vdb_result_categories.argtypes = [C.c_int64, C.c_char_p, C.c_int64]
        _H = H
    return _H


def _ck(rc):
    if rc < 0:
        msg = _host().gh_last_error().decode()
        raise _ERR.get(int(rc), ValueError if rc == _INVALID else RuntimeError)(msg)
    return rc


def _ms(t):
    if t is None:
        return 0
    if isinstance(t, dt.datetime):
        if t.tzinfo is None:
            t = t.replace(tzinfo=dt.timezone.utc)
        return int(round(t.timestamp() * 1000))
    return int(t)


class Vector:
    """vectors.Vector (database.go:90-97); Timestamp in milliseconds since the epoch (what the backends store)."""

    def __init__(self, Id="", Values=(), Indices=(), IsHidden=False, Categories=(), Timestamp=0):
        self.Id, self.Values, self.Indices = Id, [float(np.float32(v)) for v in Values], [int(i) for i in Indices]
        self.IsHidden, self.Categories, self.Timestamp = bool(IsHidden), list(Categories), _ms(Timestamp)

    def __eq__(self, o):
        return (self.Id, self.Values, self.Indices, self.IsHidden, self.Categories, self.Timestamp) == \
               (o.Id, o.Values, o.Indices, o.IsHidden, o.Categories, o.Timestamp)

    def __repr__(self):
        return "Vector(%r, %r, hidden=%r, cats=%r, ts=%r)" % (self.Id, self.Values, self.IsHidden, self.Categories, self.Timestamp)


class ScoredVector(Vector):
    Score = 0.0


class Database:
    """vectors.Database.  Methods and error behaviour follow database.go:107-120 / xvec.go."""

    def __init__(self, url="hip://", searcher=None, sparse_searcher=None):
        H = _host()
        if searcher is not None or sparse_searcher is not None:
            self._cb = SEARCH_CB(searcher) if searcher is not None else None  # keep the thunks alive
            self._scb = SPARSE_CB(sparse_searcher) if sparse_searcher is not None else None
            self.h = C.c_void_p(H.gh_vdb_open_with_searchers(C.cast(self._cb, C.c_void_p) if self._cb else None,
                                                             C.cast(self._scb, C.c_void_p) if self._scb else None))
        else:
            p = H.gh_vdb_open(url.encode())
            if not p:
                raise RuntimeError(H.gh_last_error().decode())
            self.h = C.c_void_p(p)

    def __del__(self):
        if getattr(self, "h", None):
            _host().gh_vdb_free(self.h)
            self.h = None

    def Init(self):
        return None

    def Optimize(self, name):
        return None

    def Close(self):
        _ck(_host().gh_vdb_close(self.h))

    def ListCollections(self):
        H = _host()
        n = _ck(H.gh_vdb_list(self.h, None, 0))
        buf = C.create_string_buffer(int(n))
        _ck(H.gh_vdb_list(self.h, buf, n))
        s = buf.value.decode()
        return s.split("\n") if s else []

    def DescribeCollection(self, name):
        d, dist, bits = C.c_int32(), C.c_int32(), C.c_int32()
        _ck(_host().gh_vdb_describe(self.h, name.encode(), C.byref(d), C.byref(dist), C.byref(bits)))
        return {"Name": name, "Dimension": d.value, "Distance": dist.value, "Type": "", "Bits": bits.value}

    def AddCollection(self, name, dimensions, distance, quantization="", bits=0):
        _ck(_host().gh_vdb_add_collection(self.h, name.encode(), dimensions, distance, quantization.encode(), bits))

    def DeleteCollection(self, name):
        _ck(_host().gh_vdb_delete_collection(self.h, name.encode()))

    def CountVectors(self, name):
        return int(_ck(_host().gh_vdb_count(self.h, name.encode())))

    @staticmethod
    def _stage(v):
        vals = np.ascontiguousarray(v.Values, np.float32)
        ind = np.ascontiguousarray(v.Indices, np.uint32)
        _host().gh_vdb_stage_vector(v.Id.encode(), vals.ctypes.data_as(C.POINTER(C.c_float)), vals.size,
                                    ind.ctypes.data_as(C.POINTER(C.c_uint32)), ind.size, int(v.IsHidden),
                                    "\n".join(v.Categories).encode(), v.Timestamp)

    @staticmethod
    def _results(scored):
        H = _host()
        out = []
        for r in range(int(H.gh_vdb_result_count())):
            v = ScoredVector() if scored else Vector()
            v.Id = H.gh_vdb_result_id(r).decode()
            vals = np.empty(H.gh_vdb_result_dim(r), np.float32)
            H.gh_vdb_result_values(r, vals.ctypes.data_as(C.POINTER(C.c_float)))
            v.Values = [float(x) for x in vals]
            ind = np.empty(H.gh_vdb_result_nnz(r), np.uint32)
            H.gh_vdb_result_indices(r, ind.ctypes.data_as(C.POINTER(C.c_uint32)))
            v.Indices = [int(x) for x in ind]
            v.IsHidden = bool(H.gh_vdb_result_hidden(r))
            n = H.gh_vdb_result_categories(r, None, 0)
            buf = C.create_string_buffer(int(n))
            H.gh_vdb_result_categories(r, buf, n)
            s = buf.value.decode()
            v.Categories = s.split("\n") if s else []
            v.Timestamp = int(H.gh_vdb_result_timestamp(r))
            if scored:
                v.Score = float(H.gh_vdb_result_score(r))
            out.append(v)
        return out

    def AddVectors(self, name, vectors):
        H = _host()
        H.gh_vdb_stage_clear()
        for v in vectors:
            self._stage(v)
        _ck(H.gh_vdb_add_staged(self.h, name.encode()))

    def GetVectors(self, name, ids):
        _ck(_host().gh_vdb_get(self.h, name.encode(), "\n".join(ids or []).encode()))
        return self._results(False)

    def DeleteVectors(self, name, timestamp):
        _ck(_host().gh_vdb_delete_vectors(self.h, name.encode(), _ms(timestamp)))

    def QueryVectors(self, name, q, categories, topK):
        H = _host()
        H.gh_vdb_stage_clear()
        self._stage(q)
        _ck(H.gh_vdb_query_staged(self.h, name.encode(), "\n".join(categories or []).encode(), topK))
        return self._results(True)

    def QueryVectorsBatch(self, name, Q, categories, topK):
        """Bulk form: rows of Q are dense queries; one device search per over-fetch round (gorse_vectors.hpp)."""
        H = _host()
        Q = np.ascontiguousarray(Q, np.float32)
        _ck(H.gh_vdb_query_batch(self.h, name.encode(), Q.ctypes.data_as(C.POINTER(C.c_float)), Q.shape[0], Q.shape[1],
                                 "\n".join(categories or []).encode(), topK))
        flat = self._results(True)
        cuts = [int(H.gh_vdb_result_split(t)) for t in range(Q.shape[0] + 1)]
        return [flat[cuts[t]:cuts[t + 1]] for t in range(Q.shape[0])]


    def QuerySparseBatch(self, name, queries, categories, topK):
        """Bulk form for a sparse collection: every Vector of `queries` (Indices / Values) in one device search."""
        H = _host()
        H.gh_vdb_stage_clear()
        for q in queries:
            self._stage(q)
        _ck(H.gh_vdb_query_sparse_staged(self.h, name.encode(), "\n".join(categories or []).encode(), topK))
        flat = self._results(True)
        cuts = [int(H.gh_vdb_result_split(t)) for t in range(len(queries) + 1)]
        return [flat[cuts[t]:cuts[t + 1]] for t in range(len(queries))]


def Open(path, tablePrefix=""):
    """vectors.Open (database.go:167-175): creators by URL prefix; 'hip://' is the one registered here."""
    return Database(url=path)


# ---- logics: item-to-item / user-to-user, embedding and sparse kinds (logics/item_to_item.go, user_to_user.go, vector_writer.go) ----
def ItemToItemCollection(name):
    return "item_to_item_" + name  # database.go:56-58


def UserToUserCollection(name):
    return "user_to_user_" + name  # database.go:60-62


def _logics_host():
    H = _host()
    if not getattr(H, "_logics_ready", False):
        H.gh_vwriter_new.restype = C.c_void_p
        H.gh_vwriter_new.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int64, C.c_int32]
        H.gh_vwriter_new_sparse.restype = C.c_void_p
        H.gh_vwriter_new_sparse.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int32]
        H.gh_logics_stage_kind_vector.restype = None
        H.gh_logics_stage_kind_vector.argtypes = [C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_int64, C.POINTER(C.c_int32),
                                                  C.c_int32, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                                  C.POINTER(C.c_float), C.c_int32]
        H.gh_logics_query_similar_typed.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32]
        H.gh_logics_query_similar_typed_bulk.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32]
        H.gh_vwriter_free.argtypes = [C.c_void_p]
        H.gh_vwriter_add_staged.argtypes = [C.c_void_p]
        H.gh_vwriter_clean.argtypes = [C.c_void_p]
        H.gh_logics_query_similar.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32]
        H.gh_logics_query_similar_bulk.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32]
        H.gh_logics_cf_recommend_bulk.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int64, C.c_int32, C.c_char_p,
                                                  C.c_int32]
        H.gh_logics_result_score.restype = C.c_double
        H.gh_logics_result_score.argtypes = [C.c_int64]
        H._logics_ready = True
    return H


class Score:
    """cache.Score as QueryItemToItem fills it"""

    def __init__(self, Id, Score, Categories):
        self.Id, self.Score, self.Categories = Id, Score, Categories


class EmbeddingItemToItem:
    """embeddingItemToItem (logics/item_to_item.go:122-152) = a dense VectorWriter with distance Euclidean.  `Add` takes
    the already extracted embedding (the expr column evaluation of the reference is configuration plumbing)."""

    def __init__(self, name, timestamp, client, batch_size=0, collection=None):
        self.client, self.timestamp = client, _ms(timestamp)
        self.collection = collection or ItemToItemCollection(name)
        self.w = C.c_void_p(_logics_host().gh_vwriter_new(client.h, self.collection.encode(), Euclidean, self.timestamp, batch_size))

    def __del__(self):
        if getattr(self, "w", None):
            _logics_host().gh_vwriter_free(self.w)
            self.w = None

    def Add(self, item_id, embedding, is_hidden=False, categories=()):
        if embedding is None or isinstance(embedding, str) or len(embedding) == 0:
            return None  # ExtractItemEmbedding failed / empty (item_to_item.go:137-141): nothing is written
        H = _logics_host()
        H.gh_vdb_stage_clear()
        Database._stage(Vector(item_id, embedding, IsHidden=is_hidden, Categories=categories, Timestamp=self.timestamp))
        _ck(H.gh_vwriter_add_staged(self.w))

    def Clean(self):
        _ck(_logics_host().gh_vwriter_clean(self.w))


class SparseSimilarity:
    """The sparse kinds of NewItemToItem / NewUserToUser (logics/item_to_item.go:89-113, 169-245; user_to_user.go:89-113,
    160-237): kind "tags" (label ids, tags IDF), "users" / "items" (feedback ids, their IDF), "auto" (both, feedback ids
    offset by len(tagsIDF)).  One class serves items and users; `Add` takes the label ids the column expression yields."""
    KINDS = {"tags": 0, "users": 1, "items": 1, "auto": 2}

    def __init__(self, kind, collection, timestamp, client, tags_idf=None, feedback_idf=None, batch_size=0):
        if kind not in self.KINDS:
            raise ValueError("invalid item-to-item type")  # item_to_item.go:110-112
        if kind in ("tags", "auto") and tags_idf is None:
            raise ValueError("tags IDF is required for %s" % kind)  # :97-99, 106-108
        if kind in ("users", "items", "auto") and feedback_idf is None:
            raise ValueError("%s IDF is required for %s" % ("users" if kind != "items" else "items", kind))
        self.kind, self.collection, self.client, self.timestamp = kind, collection, client, _ms(timestamp)
        self.tags_idf = np.ascontiguousarray(tags_idf if tags_idf is not None else [], np.float32)
        self.fb_idf = np.ascontiguousarray(feedback_idf if feedback_idf is not None else [], np.float32)
        self.w = C.c_void_p(_logics_host().gh_vwriter_new_sparse(client.h, collection.encode(), self.timestamp, batch_size))

    def __del__(self):
        if getattr(self, "w", None):
            _logics_host().gh_vwriter_free(self.w)
            self.w = None

    def Add(self, id, tags=(), feedback=(), is_hidden=False, categories=()):
        H = _logics_host()
        t = np.ascontiguousarray(list(tags), np.int32)
        f = np.ascontiguousarray(list(feedback), np.int32)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        H.gh_vdb_stage_clear()
        H.gh_logics_stage_kind_vector(self.KINDS[self.kind], str(id).encode(), int(is_hidden), "\n".join(categories).encode(),
                                      self.timestamp, t.ctypes.data_as(ip), t.size, self.tags_idf.ctypes.data_as(fp),
                                      self.tags_idf.size, f.ctypes.data_as(ip), f.size, self.fb_idf.ctypes.data_as(fp),
                                      self.fb_idf.size)
        _ck(H.gh_vwriter_add_staged(self.w))

    def Clean(self):
        _ck(_logics_host().gh_vwriter_clean(self.w))


def QuerySimilarTyped(client, collection, kind, id, categories, n):
    """QueryItemToItem / QueryUserToUser for any kind (item_to_item.go:50-88): Dot unless "embedding", the id itself and
    scores <= 0 skipped, "auto" halved"""
    _ck(_logics_host().gh_logics_query_similar_typed(client.h, collection.encode(), kind.encode(), str(id).encode(),
                                                      "\n".join(categories or []).encode(), n))
    return _scores()


def QuerySimilarTypedBulk(client, collection, kind, ids, categories, n):
    """a sparse kind's neighbours for many ids with ONE device search (logics::QuerySimilarTypedBulk)"""
    H = _logics_host()
    _ck(H.gh_logics_query_similar_typed_bulk(client.h, collection.encode(), kind.encode(), "\n".join(map(str, ids)).encode(),
                                             "\n".join(categories or []).encode(), n))
    flat = _scores()
    cuts = [int(H.gh_vdb_result_split(t)) for t in range(len(ids) + 1)]
    return [flat[cuts[t]:cuts[t + 1]] for t in range(len(ids))]


def _scores():
    H = _logics_host()
    out = []
    for r, v in enumerate(Database._results(False)):
        out.append(Score(v.Id, float(H.gh_logics_result_score(r)), v.Categories))
    return out


def QuerySimilar(client, collection, item_id, categories, n):
    """QueryItemToItem / QueryUserToUser for the embedding type (item_to_item.go:50-88)"""
    _ck(_logics_host().gh_logics_query_similar(client.h, collection.encode(), item_id.encode(),
                                                "\n".join(categories or []).encode(), n))
    return _scores()


def QuerySimilarBulk(client, collection, ids, categories, n):
    """every id's neighbours with one bulk device search (gorse_vectors.hpp logics::QuerySimilarBulk)"""
    H = _logics_host()
    _ck(H.gh_logics_query_similar_bulk(client.h, collection.encode(), "\n".join(ids).encode(),
                                       "\n".join(categories or []).encode(), n))
    flat = _scores()
    cuts = [int(H.gh_vdb_result_split(t)) for t in range(len(ids) + 1)]
    return [flat[cuts[t]:cuts[t + 1]] for t in range(len(ids))]


def CollaborativeFilteringCollection(model_id):
    return "collaborative_filtering_%d" % model_id  # database.go:52-54


def CollaborativeRecommendBulk(client, collection, embeddings, excludes, cache_size):
    """updateCollaborativeRecommend (worker/pipeline.go:403-425) for many users with one bulk device search; excludes[t] =
    the ids user t has already seen.  Returns one Score list per user."""
    H = _logics_host()
    Q = np.ascontiguousarray(embeddings, np.float32)
    ex = "\x1e".join("\n".join(e) for e in excludes)
    _ck(H.gh_logics_cf_recommend_bulk(client.h, collection.encode(), Q.ctypes.data_as(C.POINTER(C.c_float)), Q.shape[0],
                                      Q.shape[1], ex.encode(), cache_size))
    flat = _scores()
    cuts = [int(H.gh_vdb_result_split(t)) for t in range(Q.shape[0] + 1)]
    return [flat[cuts[t]:cuts[t + 1]] for t in range(Q.shape[0])]


class MatrixFactorizationItems:
    """logics.MatrixFactorizationItems (logics/cf.go:36-128): item id -> embedding, nearest items by -dot on the exact index, and
    its blob: reads the reference's HNSW-based stream (keeping the vectors and ids) and this library's own versioned framing,
    writes the latter.  searcher = a search callback (the CPU test-suite) or None for the GPU."""

    def __init__(self, timestamp_unix_nanos=0, searcher=None):
        H = _host()
        H.gh_mfitems_new.restype = C.c_void_p
        H.gh_mfitems_new.argtypes = [C.c_int64, C.c_void_p]
        for n, r, a in (("gh_mfitems_free", None, [C.c_void_p]), ("gh_mfitems_add", None, [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int32]),
                        ("gh_mfitems_count", C.c_int64, [C.c_void_p]), ("gh_mfitems_dimension", C.c_int32, [C.c_void_p]),
                        ("gh_mfitems_timestamp", C.c_int64, [C.c_void_p]), ("gh_mfitems_id", C.c_int64, [C.c_void_p, C.c_int64, C.c_char_p, C.c_int64]),
                        ("gh_mfitems_row", None, [C.c_void_p, C.c_int64, C.POINTER(C.c_float)]),
                        ("gh_mfitems_marshal", C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
                        ("gh_mfitems_marshal_reference", C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
                        ("gh_mfitems_unmarshal", C.c_int32, [C.c_void_p, C.c_char_p, C.c_int64]),
                        ("gh_mfitems_search", C.c_int64, [C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_int32, C.c_char_p, C.c_int64,
                                                          C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int32)])):
            getattr(H, n).restype = r
            getattr(H, n).argtypes = a
        self._cb = SEARCH_CB(searcher) if searcher is not None else None
        self.p = C.c_void_p(H.gh_mfitems_new(int(timestamp_unix_nanos), C.cast(self._cb, C.c_void_p) if self._cb else None))

    def __del__(self):
        if getattr(self, "p", None):
            _host().gh_mfitems_free(self.p)
            self.p = None

    def Add(self, item_id, v):
        v = np.ascontiguousarray(v, np.float32)
        _host().gh_mfitems_add(self.p, str(item_id).encode(), v.ctypes.data_as(C.POINTER(C.c_float)), v.size)

    def Count(self):
        return _host().gh_mfitems_count(self.p)

    def Dimension(self):
        return _host().gh_mfitems_dimension(self.p)

    def Timestamp(self):
        return _host().gh_mfitems_timestamp(self.p)

    def Id(self, i):
        buf = C.create_string_buffer(4096)
        n = _host().gh_mfitems_id(self.p, i, buf, len(buf))
        return buf.raw[:n].decode()

    def Row(self, i):
        out = np.empty(self.Dimension(), np.float32)
        _host().gh_mfitems_row(self.p, i, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def Marshal(self):
        n = _host().gh_mfitems_marshal(self.p, None, 0)
        buf = C.create_string_buffer(max(int(n), 1))
        _host().gh_mfitems_marshal(self.p, buf, len(buf))
        return buf.raw[:n]

    def MarshalReference(self):
        """the blob in the reference's own format (HNSW.Marshal) with a graph built by the device's exact all-pairs search: what a
        master with this library writes for workers without it (gorse_vectors.hpp MarshalReference)"""
        n = _host().gh_mfitems_marshal_reference(self.p, None, 0)
        if n < 0:
            raise RuntimeError("MarshalReference failed: %s" % _host().gh_last_error().decode("utf-8", "replace"))
        buf = C.create_string_buffer(max(int(n), 1))
        _host().gh_mfitems_marshal_reference(self.p, buf, len(buf))
        return buf.raw[:n]

    def Unmarshal(self, blob):
        _ck(_host().gh_mfitems_unmarshal(self.p, blob, len(blob)))

    def Search(self, v, n):
        v = np.ascontiguousarray(v, np.float32)
        ids = C.create_string_buffer(1 << 16)
        scores = (C.c_double * max(n, 1))()
        cnt = C.c_int32(0)
        nb = _host().gh_mfitems_search(self.p, v.ctypes.data_as(C.POINTER(C.c_float)), v.size, n, ids, len(ids), scores, n, C.byref(cnt))
        if nb < 0:
            _ck(-1)
        names = ids.raw[:nb].decode().split("\n") if cnt.value else []
        return [(names[t], scores[t]) for t in range(cnt.value)]


class MatrixFactorizationUsers:
    """logics.MatrixFactorizationUsers (logics/cf.go:122-179): user id -> embedding, the blob workers download"""

    def __init__(self, ptr=None):
        H = _logics_host()
        H.gh_mfusers_new.restype = C.c_void_p
        H.gh_mfusers_free.argtypes = [C.c_void_p]
        H.gh_mfusers_add.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int32]
        H.gh_mfusers_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int32]
        H.gh_mfusers_count.argtypes = [C.c_void_p]
        H.gh_mfusers_marshal.restype = C.c_int64
        H.gh_mfusers_marshal.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        H.gh_mfusers_unmarshal.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        self.p = C.c_void_p(ptr if ptr is not None else H.gh_mfusers_new())

    def __del__(self):
        if getattr(self, "p", None):
            _logics_host().gh_mfusers_free(self.p)
            self.p = None

    def Add(self, user_id, v):
        a = np.ascontiguousarray(v, np.float32)
        _logics_host().gh_mfusers_add(self.p, str(user_id).encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size)

    def Get(self, user_id, max_dim=4096):
        out = np.zeros(max_dim, np.float32)
        n = _logics_host().gh_mfusers_get(self.p, str(user_id).encode(), out.ctypes.data_as(C.POINTER(C.c_float)), max_dim)
        return (out[:n].copy(), True) if n >= 0 else (None, False)

    def Count(self):
        return _logics_host().gh_mfusers_count(self.p)

    def Marshal(self):
        H = _logics_host()
        n = H.gh_mfusers_marshal(self.p, None, 0)
        buf = C.create_string_buffer(int(n))
        H.gh_mfusers_marshal(self.p, buf, n)
        return buf.raw

    def Unmarshal(self, blob):
        _ck(_logics_host().gh_mfusers_unmarshal(self.p, blob, len(blob)))


def PublishCollaborativeFiltering(model, client, model_id, hidden=None, categories=None, batch_size=0):
    """What the master does with a fitted model (master/tasks.go:925-969): the predictable items' factors into the Dot
    collection collaborative_filtering_<model_id>, the predictable users' factors into a MatrixFactorizationUsers."""
    H = _logics_host()
    H.gh_publish_cf.restype = C.c_void_p
    H.gh_publish_cf.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]
    hid = None if hidden is None else np.ascontiguousarray(hidden, np.uint8).tobytes()
    cats = None if categories is None else "\x1e".join("\n".join(c) for c in categories).encode()
    ptr = H.gh_publish_cf(model.p, client.h, int(model_id), hid, 0 if hidden is None else len(hidden), cats, batch_size)
    if not ptr:
        raise RuntimeError(H.gh_last_error().decode("utf-8", "replace"))
    return MatrixFactorizationUsers(ptr)
