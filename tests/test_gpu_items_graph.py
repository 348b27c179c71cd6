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
    assert len(bottom) == n and all(len(q) == 96 for _, q in bottom) and len(upper) >= 1
    # spot checks of the graph against plain numpy: the bottom queue of a vector = its 96 nearest others, ascending
    for i in rng.choice(n, 20, replace=False):
        dist = -(X @ X[i])
        dist[i] = np.inf
        want = set(np.argsort(dist, kind="stable")[:96].tolist())
        got = [v for v, _ in bottom[i][1]]
        assert len(set(got) & want) >= 94  # ties at the cut may differ from numpy's float order
        w = [x for _, x in bottom[i][1]]
        assert w == sorted(w)
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
