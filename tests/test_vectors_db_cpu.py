"""vectors.Database twin (gorse_amd/host/gorse_vectors.hpp) against the reference's own test-suite, with the device searches
replaced by exact checkers built on the oracle (ann.Bruteforce and the sparse inner-product ranking restated): collection / vector bookkeeping, filters,
over-fetch logic and score conventions are host code and are fully exercised here; test_gpu_vectors_db.py runs the same
suite on the MI355X."""
import numpy as np
import pytest

import vectors_suite as S
from gorse_amd import vectors as V
from oracle import oracle as orc


@pytest.fixture
def db(oracle):
    oracle.set_isa(orc.ISA_AVX512)

    def searcher(X, n, d, metric, Q, nq, k, idx, dist, cnt):
        Xa = np.ctypeslib.as_array(X, (n, d)).copy()
        Qa = np.ctypeslib.as_array(Q, (nq, d)).copy()
        I, D, Cn = np.ctypeslib.as_array(idx, (nq, k)), np.ctypeslib.as_array(dist, (nq, k)), np.ctypeslib.as_array(cnt, (nq,))
        for t in range(nq):
            ei, ed = oracle.search_vector(Xa, metric, Qa[t], k)
            I[t, :ei.size], D[t, :ei.size], Cn[t] = ei, ed, ei.size
        return 0

    def sparse_searcher(n, indptr, indices, values, admissible, nq, q_indptr, q_indices, q_values, k, idx, score, cnt):
        ptr = np.ctypeslib.as_array(indptr, (n + 1,)).copy()
        nnz = int(ptr[-1])
        ind = np.ctypeslib.as_array(indices, (max(nnz, 1),))[:nnz].copy()
        val = np.ctypeslib.as_array(values, (max(nnz, 1),))[:nnz].copy()
        ok = np.ctypeslib.as_array(admissible, (n,)).copy()
        qp = np.ctypeslib.as_array(q_indptr, (nq + 1,)).copy()
        qn = int(qp[-1])
        qi = np.ctypeslib.as_array(q_indices, (max(qn, 1),))[:qn].copy()
        qv = np.ctypeslib.as_array(q_values, (max(qn, 1),))[:qn].copy()
        I, Sc, Cn = np.ctypeslib.as_array(idx, (nq, k)), np.ctypeslib.as_array(score, (nq, k)), np.ctypeslib.as_array(cnt, (nq,))
        for t in range(nq):
            ei, es = oracle.sparse_search(ptr, ind, val, qi[qp[t]:qp[t + 1]], qv[qp[t]:qp[t + 1]], k, admissible=ok)
            I[t, :ei.size], Sc[t, :ei.size], Cn[t] = ei, es, ei.size
        return 0
    return V.Database(searcher=searcher, sparse_searcher=sparse_searcher)


@pytest.mark.parametrize("case", [S.collections, S.vectors, S.get_vectors, S.hidden, S.dot, S.delete_vectors,
                                  S.upsert_and_close, S.item_to_item_column, S.item_to_item_embedding, S.item_to_item_clean,
                                  S.user_to_user_embedding, S.user_to_user_clean, S.collaborative_recommend] + S.SPARSE_CASES,
                         ids=lambda f: f.__name__)
def test_reference_suite(db, case):
    case(db)


@pytest.mark.parametrize("distance,name,metric", [(V.Dot, "dot", orc.METRIC_NEG_DOT), (V.Euclidean, "l2", orc.METRIC_EUCLIDEAN),
                                                  (V.Cosine, "cos", orc.METRIC_COSINE)])
def test_exact_filtered_topk(db, oracle, distance, name, metric):
    S.exact_filtered_topk(db, distance, name, lambda X, q: np.array([oracle.distance(metric, q, x) for x in X], np.float32))


def test_unknown_prefix():
    with pytest.raises(RuntimeError):
        V.Open("nosuch://")
