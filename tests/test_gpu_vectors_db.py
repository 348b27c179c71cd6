"""The reference's vectors.Database test-suite (tests/vectors_suite.py) on the MI355X: `vectors.Open("hip://")`, every
search an exact gorse_topk search on the device (literal scan for single queries, MFMA sweep for the bulk form)."""
import numpy as np
import pytest

import vectors_suite as S
from gorse_amd import vectors as V
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture
def db():
    return V.Open("hip://")


# S.collaborative_recommend (bulk recommendations for many users), S.PENDING_DENSE_CASES and the sparse collections
# (S.SPARSE_CASES) were written after the round's GPU budget was spent: they run in the CPU module against the oracle-backed
# searchers and, on the device, from tests/test_gpu_vectors_sparse.py (isolated in a child process until a device session
# has seen them green); the dense ones join this list then.


@pytest.mark.parametrize("case", [S.collections, S.vectors, S.get_vectors, S.hidden, S.dot, S.delete_vectors,
                                  S.upsert_and_close, S.item_to_item_column, S.item_to_item_embedding, S.item_to_item_clean,
                                  S.user_to_user_embedding, S.user_to_user_clean],
                         ids=lambda f: f.__name__)
def test_reference_suite(db, case):
    case(db)


@pytest.mark.parametrize("distance,name,metric", [(V.Dot, "dot", orc.METRIC_NEG_DOT), (V.Euclidean, "l2", orc.METRIC_EUCLIDEAN),
                                                  (V.Cosine, "cos", orc.METRIC_COSINE)])
def test_exact_filtered_topk(db, oracle, distance, name, metric):
    oracle.set_isa(orc.ISA_AVX512)
    S.exact_filtered_topk(db, distance, name, lambda X, q: np.array([oracle.distance(metric, q, x) for x in X], np.float32))
