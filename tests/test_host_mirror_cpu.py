"""Host-side logic of the C++ mirror that needs no GPU: heap containers and ranking metrics against the
reference's known-answer tests, dataset bookkeeping, model file framing."""
import json
import os

import numpy as np
import pytest

from gorse_amd import cf

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_topk_filter_kats():
    # common/heap/filter_test.go:22-45
    for k in KATS["heap"]["topk_filter"]:
        f = cf.TopKFilter(k["k"])
        for it, w in zip(k["items"], k["weights"]):
            f.Push(it, w)
        got = f.PopAll()
        assert [v for v, _ in got] == k["values"] and [w for _, w in got] == k["out_weights"]


def test_priority_queue_kat():
    # common/heap/pq_test.go:28-61 (pop ascending; Reverse() pops descending)
    pq = KATS["heap"]["priority_queue"]
    e = pq["elements"]
    assert [v for v, _ in cf.priority_queue_drain(False, e, e)] == pq["asc"]
    assert [v for v, _ in cf.priority_queue_drain(False, e, e, reverse=True)] == pq["desc"]
    with pytest.raises(cf.HostError):  # "NaN weight is forbidden" (pq.go:69-71)
        cf.priority_queue_drain(False, [1], [float("nan")])
    # duplicates are ignored (pq.go:72)
    assert cf.priority_queue_drain(False, [4, 4, 2], [4, 9, 2]) == [(2, 2.0), (4, 4.0)]


def test_heap_matches_oracle_on_ties(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        n, k = int(rng.integers(1, 60)), int(rng.integers(1, 12))
        items = rng.permutation(1000)[:n].astype(np.int32)
        w = rng.integers(0, 4, n).astype(np.float32)  # many ties
        f = cf.TopKFilter(k)
        for a, b in zip(items, w):
            f.Push(int(a), float(b))
        ev, ew = oracle.topk_filter(k, items, w)
        assert f.PopAll() == [(int(a), float(b)) for a, b in zip(ev, ew)]


def test_metric_kats():
    # model/cf/evaluator_test.go:33-74
    m = KATS["metrics"]
    ids = {"ndcg": cf.NDCG, "precision": cf.Precision, "recall": cf.Recall, "hr": cf.HR, "map": cf.MAP, "mrr": cf.MRR}
    for c in m["cases"]:
        assert abs(cf.metric(ids[c["metric"]], c["target"], m["rank"]) - c["out"]) < m["epsilon"]


def test_dataset_bookkeeping():
    # dataset/dataset_test.go style: AddFeedback builds both directions in insertion order
    d = cf.Dataset()
    for u, i in [("a", "x"), ("a", "y"), ("b", "x"), ("c", "z")]:
        d.AddFeedback(u, i)
    assert (d.CountUsers(), d.CountItems(), d.CountFeedback()) == (3, 3, 4)
    d.AddUser("lonely")
    assert d.CountUsers() == 4 and d.CountFeedback() == 4
