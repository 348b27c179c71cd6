"""MatrixFactorizationItems.MarshalReference on the device (gorse_amd/host/gorse_vectors.hpp): the blob a master WITH this
library writes for workers without it -- the reference's own format (logics/cf.go:81-101 around HNSW.Marshal, common/ann/
hnsw.go:278-337) with a graph built from one exact all-pairs search per layer instead of a million insertions.  The
reference's own search (hnsw.go:100-114, 187-229, restated in oracle.hnsw_knn_search) must find the true nearest items on it."""
import struct

import numpy as np
import pytest

from gorse_amd import vectors as V
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _index_section_start(blob):
    at = 0
    for _ in range(2):  # timestamp, dimension
        (ln,) = struct.unpack_from("<i", blob, at)
        at += 4 + ln
    return at


@pytest.mark.parametrize("n,d,scaled", [(20000, 32, False), (6000, 64, True)])
def test_device_built_graph_is_walked_by_the_reference_search(n, d, scaled):
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d)).astype(np.float32)
    if scaled:  # item factors of unequal length, as a trained model has them: -dot favours the long ones
        X *= rng.uniform(0.2, 2.0, (n, 1)).astype(np.float32)
    m = V.MatrixFactorizationItems(timestamp_unix_nanos=7)
    for i in range(n):
        m.Add(str(i), X[i])
    blob = m.MarshalReference()
    params, streams, bottom, upper, enter, end = orc.hnsw_parse_index_section(blob, _index_section_start(blob))
    assert len(bottom) == n and all(84 <= len(q) <= 96 for _, q in bottom) and len(upper) >= 1
    # spot checks of the graph against plain numpy: the bottom queue of a vector holds its 84 nearest others (+ reverse links), ascending
    for i in rng.choice(n, 20, replace=False):
        dist = -(X @ X[i])
        dist[i] = np.inf
        want = set(np.argsort(dist, kind="stable")[:84].tolist())
        got = [v for v, _ in bottom[i][1]]
        assert len(set(got) & want) >= 82  # ties at the cut may differ from numpy's float order
        w = [x for _, x in bottom[i][1]]
        assert w == sorted(w)
    indeg = np.zeros(n, np.int64)
    for _, q in bottom:
        for v, _ in q:
            indeg[v] += 1
    unreachable = int((indeg == 0).sum())
    print("bottom layer: %d of %d vectors without an incoming link" % (unreachable, n))
    assert unreachable <= n // 200
    k, hits, total = 10, 0, 0
    for t in range(150):
        q = rng.standard_normal(d).astype(np.float32)
        got = orc.hnsw_knn_search(X, bottom, upper, enter, params, q, k)
        want = np.argsort(-(X @ q), kind="stable")[:k]
        hits += len(set(got) & set(want.tolist()))
        total += k
    recall = hits / total
    print("device-built HNSW stream, %d x %d%s: recall@%d of the reference's search = %.3f, %d layers above the bottom"
          % (n, d, " (unequal norms)" if scaled else "", k, recall, len(upper)))
    assert recall >= 0.9
    m2 = V.MatrixFactorizationItems()
    m2.Unmarshal(blob)  # this library reads the file like any reference file
    assert m2.Count() == n and np.array_equal(m2.Row(n - 1), X[n - 1])


def test_device_built_graph_on_clustered_factors_of_unequal_length():
    """What a trained MF model's item factors look like: clusters (genres) whose members share a direction, lengths spread over an
    order of magnitude (popular items are long).  Under -dot every exact neighbour list points at the long vectors of a cluster; the
    reverse-link slots of MarshalReference keep the short ones attached.  Queries = user-like vectors (a cluster direction + noise):
    the reference's search (restated) on the device-built graph must find the true top 10, and (nearly) no vector may be without
    an incoming link."""
    rng = np.random.default_rng(77)
    n, d, nc = 12000, 32, 24
    centers = rng.standard_normal((nc, d)).astype(np.float32)
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    member = rng.integers(0, nc, n)
    X = (centers[member] + 0.35 * rng.standard_normal((n, d))).astype(np.float32)
    X *= np.exp(rng.normal(0.0, 0.8, (n, 1))).astype(np.float32)  # log-normal lengths: a factor of ~10 between short and long
    m = V.MatrixFactorizationItems(timestamp_unix_nanos=9)
    for i in range(n):
        m.Add(str(i), X[i])
    blob = m.MarshalReference()
    params, streams, bottom, upper, enter, end = orc.hnsw_parse_index_section(blob, _index_section_start(blob))
    indeg = np.zeros(n, np.int64)
    for _, q in bottom:
        for v, _ in q:
            indeg[v] += 1
    unreachable = int((indeg == 0).sum())
    k, hits, total = 10, 0, 0
    for t in range(200):
        q = (centers[t % nc] + 0.3 * rng.standard_normal(d)).astype(np.float32)
        got = orc.hnsw_knn_search(X, bottom, upper, enter, params, q, k)
        want = np.argsort(-(X @ q), kind="stable")[:k]
        hits += len(set(got) & set(want.tolist()))
        total += k
    recall = hits / total
    print("device-built HNSW stream, %d x %d clustered, log-normal lengths: recall@%d of the reference's search = %.3f; %d vectors without "
          "an incoming link (median in-degree %d, of the shortest tenth %d)"
          % (n, d, k, recall, unreachable, np.median(indeg), np.median(indeg[np.argsort(np.linalg.norm(X, axis=1))[:n // 10]])))
    assert recall >= 0.9
    assert unreachable <= n // 200
