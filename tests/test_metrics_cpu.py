"""gorse_amd.metrics (the host side of Evaluate) against the reference's known answers (model/cf/evaluator_test.go:31-74,
transcribed in tests/golden/reference_kats.json) and against the oracle's restatement on random inputs."""
import json
import os

import numpy as np
import pytest

from gorse_amd import metrics as M
from oracle import oracle as orc

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_reference_known_answers():
    k = KATS["metrics"]
    for case in k["cases"]:
        got = M.METRICS[case["metric"]](case["target"], k["rank"])
        assert abs(float(got) - case["out"]) < k["epsilon"], case


@pytest.mark.parametrize("name,code", [("ndcg", orc.M_NDCG), ("precision", orc.M_PRECISION), ("recall", orc.M_RECALL),
                                       ("hr", orc.M_HR), ("map", orc.M_MAP), ("mrr", orc.M_MRR)])
def test_metrics_match_the_oracle(oracle, name, code):
    rng = np.random.default_rng(code)
    for _ in range(200):
        n_rank = int(rng.integers(1, 12))
        rank = rng.permutation(40)[:n_rank].astype(np.int32)
        target = rng.permutation(40)[:int(rng.integers(1, 8))].astype(np.int32)
        got = float(M.METRICS[name](target, rank))
        ref = oracle.metric(code, target, rank)
        assert got == pytest.approx(ref, abs=1e-6), (name, target, rank)


def test_partial_sums_add_up_to_evaluate(oracle):
    # two "workers" over disjoint user ranges: partial sums / counts added and scaled = the oracle's Evaluate
    from gorse_amd import synth
    data = synth.synth_cf(90, 60, 1500, seed=2, min_len=3, n_neg=20)
    P, Q = synth.init_factors(data.U, data.I, 8, 0.0, 0.3, 4)
    ref = oracle.evaluate(P, Q, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)
    total, count = np.zeros(3, np.float32), np.float32(0)
    for lo, hi in ((0, 37), (37, 90)):
        users = np.array([u for u in range(lo, hi) if data.test_ptr[u + 1] > data.test_ptr[u]], np.int32)
        cptr, cidx = M.candidates(data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, users)
        rank, rlen = oracle.mf_rank(P, Q, users, cptr, cidx, 10)
        s, c = M.partial_sums(rank, rlen, users, data.test_ptr, data.test_idx)
        total += s
        count += c
    assert np.allclose(total / count, ref, atol=2e-6)
