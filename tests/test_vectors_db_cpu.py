"""vectors.Database twin (gorse_amd/host/gorse_vectors.hpp) against the reference's own test-suite, with the device search
replaced by an exact checker built on the oracle (ann.Bruteforce restated): collection / vector bookkeeping, filters,
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
    return V.Database(searcher=searcher)


@pytest.mark.parametrize("case", [S.collections, S.vectors, S.get_vectors, S.sparse, S.hidden, S.dot, S.delete_vectors,
                                  S.upsert_and_close, S.item_to_item_column, S.item_to_item_embedding, S.item_to_item_clean,
                                  S.user_to_user_embedding, S.user_to_user_clean, S.collaborative_recommend],
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
