"""The reference's vectors.Database test-suite (storage/vectors/database_test.go:33-330), transcribed; the two test
modules run it against the C++ twin with the GPU searcher (test_gpu_vectors_db.py) and with an exact CPU checker built on
the oracle (test_vectors_db_cpu.py).  defaultVectorSize and the vectors are the reference's."""
import numpy as np
import pytest

from gorse_amd import vectors as V

defaultVectorSize = 4  # database_test.go: vectors of 4 floats ({1,0,0,0}, {0.9,0.1,0,0}, ...)


def vec(*xs):
    v = [0.0] * defaultVectorSize
    for i, x in enumerate(xs):
        v[i] = x
    return v


def collections(db):  # TestCollections (database_test.go:38-84)
    assert db.ListCollections() == []
    db.AddCollection("test", defaultVectorSize, V.Cosine)
    with pytest.raises(V.ErrAlreadyExists):
        db.AddCollection("test", defaultVectorSize, V.Cosine)
    info = db.DescribeCollection("test")
    assert info["Name"] == "test" and info["Dimension"] == defaultVectorSize and info["Distance"] == V.Cosine
    assert info["Type"] == "" and info["Bits"] == 0
    assert db.ListCollections() == ["test"]
    db.DeleteCollection("test")
    with pytest.raises(V.ErrNotFound):
        db.DescribeCollection("test")
    assert db.ListCollections() == []
    with pytest.raises(V.ErrNotFound):
        db.DeleteCollection("non-existent")


def vectors(db):  # TestVectors (database_test.go:86-153)
    db.AddCollection("test", defaultVectorSize, V.Cosine)
    assert db.CountVectors("test") == 0
    a, b = vec(1), vec(0.9, 0.1)
    db.AddVectors("test", [V.Vector("a", a, Categories=["cat-a", "common"]), V.Vector("b", b, Categories=["cat-b", "common"])])
    assert db.CountVectors("test") == 2
    r = db.QueryVectors("test", V.Vector(Values=a), ["cat-a"], 10)
    assert [x.Id for x in r] == ["a"] and r[0].Categories
    r = db.QueryVectors("test", V.Vector(Values=a), ["common"], 10)
    assert [x.Id for x in r] == ["a", "b"] and r[0].Score > r[1].Score and all(x.Categories for x in r)
    r = db.QueryVectors("test", V.Vector(Values=a), ["cat-a", "common"], 10)
    assert [x.Id for x in r] == ["a"]
    r = db.QueryVectors("test", V.Vector(Values=a), None, 1)
    assert r and all(x.Categories for x in r)
    q = db.GetVectors("test", ["a"])
    assert len(q) == 1
    r = db.QueryVectors("test", q[0], None, 10)  # the query vector itself is not excluded
    assert [x.Id for x in r] == ["a", "b"]
    assert db.QueryVectors("test", V.Vector(Values=a), None, 0) == []  # topK <= 0 (xvec.go:380-382)
    with pytest.raises(V.ErrNotFound):
        db.QueryVectors("nope", V.Vector(Values=a), None, 3)


def get_vectors(db):  # TestGetVectors (database_test.go:155-190)
    db.AddCollection("test", defaultVectorSize, V.Cosine)
    ta = 1_790_000_000_123
    va = V.Vector("a", [1, 0, 0, 0], Categories=["cat-a", "common"], Timestamp=ta)
    vb = V.Vector("b", [0, 1, 0, 0], IsHidden=True, Categories=["cat-b", "common"], Timestamp=ta + 1000)
    db.AddVectors("test", [va, vb])
    assert db.GetVectors("test", ["b", "missing", "a", "b"]) == [vb, va]
    assert db.GetVectors("test", None) == []


def sparse(db):  # TestSparse (database_test.go:192-243): sparse collections are the reference's Flat index; not built here
    with pytest.raises(V.ErrNotSupported):
        db.AddCollection("test_sparse", 0, V.Dot)
    db.AddCollection("dense", defaultVectorSize, V.Dot)
    with pytest.raises(V.ErrNotSupported):
        db.AddVectors("dense", [V.Vector("old", [1, 1], Indices=[1, 100])])
    with pytest.raises(V.ErrNotSupported):
        db.QueryVectors("dense", V.Vector(Values=[1, 2], Indices=[1, 100]), None, 10)
    with pytest.raises(V.ErrNotSupported):  # quantized collections (database_test.go:332-420) neither
        db.AddCollection("q", defaultVectorSize, V.Cosine, quantization="sq", bits=8)


def hidden(db):  # TestHidden (database_test.go:245-279)
    db.AddCollection("test_hidden", defaultVectorSize, V.Cosine)
    query = [1, 0, 0, 0]
    db.AddVectors("test_hidden", [V.Vector("visible", [0.9, 0.1, 0, 0], Categories=["common", "quo'te"]),
                                  V.Vector("hidden", query, IsHidden=True, Categories=["common", "quo'te"])])
    assert db.CountVectors("test_hidden") == 2
    for cats in (None, ["common"], ["quo'te"]):
        r = db.QueryVectors("test_hidden", V.Vector(Values=query), cats, 10)
        assert [x.Id for x in r] == ["visible"]


def dot(db):  # TestDot (database_test.go:281-300)
    db.AddCollection("test_dot", defaultVectorSize, V.Dot)
    db.AddVectors("test_dot", [V.Vector("a", [2, 0, 0, 0]), V.Vector("b", [1, 1, 0, 0])])
    r = db.QueryVectors("test_dot", V.Vector(Values=[1, 0, 0, 0]), None, 2)
    assert [x.Id for x in r] == ["a", "b"] and r[0].Score > r[1].Score
    assert r[0].Score == 2.0 and r[1].Score == 1.0  # Dot: the score IS the inner product


def delete_vectors(db):  # TestDeleteVectors (database_test.go:302-330)
    db.AddCollection("test", defaultVectorSize, V.Cosine)
    cutoff = 1_790_000_000_000
    db.AddVectors("test", [V.Vector("old", vec(1), Categories=["common"], Timestamp=cutoff - 3_600_000),
                           V.Vector("new", vec(0.9, 0.1), Categories=["common"], Timestamp=cutoff)])
    db.DeleteVectors("test", cutoff)
    assert db.CountVectors("test") == 1
    r = db.QueryVectors("test", V.Vector(Values=vec(1)), ["common"], 10)
    assert [x.Id for x in r] == ["new"]


def upsert_and_close(db):  # xvec.go:301-327 (Upsert), :166-185 (closed database)
    db.AddCollection("c", defaultVectorSize, V.Euclidean)
    db.AddVectors("c", [V.Vector("a", [1, 0, 0, 0]), V.Vector("b", [0, 1, 0, 0])])
    db.AddVectors("c", [V.Vector("a", [0, 0, 1, 0], Categories=["x"])])  # same Id: replaced, not appended
    assert db.CountVectors("c") == 2
    assert db.GetVectors("c", ["a"])[0].Values == [0, 0, 1, 0]
    r = db.QueryVectors("c", V.Vector(Values=[0, 0, 1, 0]), None, 2)
    assert [x.Id for x in r] == ["a", "b"] and r[0].Score == 0.0 and abs(r[1].Score + np.sqrt(2)) < 1e-6  # negated distance
    with pytest.raises(ValueError):
        db.AddVectors("c", [V.Vector("bad", [1, 2, 3])])  # wrong dimension: nothing is added
    assert db.CountVectors("c") == 2
    db.Close()
    with pytest.raises(RuntimeError):
        db.CountVectors("c")


def exact_filtered_topk(db, distance, metric_name, brute):
    """Over-fetch + filter = the exact top-K of the admissible set, one query at a time and in bulk."""
    rng = np.random.default_rng(distance + 7)
    n, d, topk = 700, 24, 12
    X = rng.standard_normal((n, d)).astype(np.float32)
    cats = [["a"] if r % 3 == 0 else (["a", "b"] if r % 3 == 1 else ["c"]) for r in range(n)]
    hid = rng.random(n) < 0.4  # many hidden vectors: the first fetch is not enough for most queries
    db.AddCollection(metric_name, d, distance)
    db.AddVectors(metric_name, [V.Vector("v%d" % r, X[r], IsHidden=bool(hid[r]), Categories=cats[r]) for r in range(n)])
    Q = rng.standard_normal((70, d)).astype(np.float32)
    for want in (None, ["a"], ["a", "b"], ["zzz"]):
        ok = np.array([not hid[r] and all(c in cats[r] for c in (want or [])) for r in range(n)])
        bulk = db.QueryVectorsBatch(metric_name, Q, want, topk)
        for t in range(Q.shape[0]):
            dist = brute(X, Q[t])
            order = [r for r in np.argsort(dist, kind="stable") if ok[r]][:topk]
            got = bulk[t]
            assert len(got) == len(order)
            # equal distances may come in either order: compare as (distance, id) with the distances exact
            assert [np.float32(-g.Score) for g in got] == [np.float32(dist[r]) for r in order], (metric_name, want, t)
            assert sorted(g.Id for g in got) == sorted("v%d" % r for r in order) or \
                len(set(np.float32(dist[r]) for r in order)) < len(order)
            assert not any(g.IsHidden for g in got)
            if t < 5:
                one = db.QueryVectors(metric_name, V.Vector(Values=Q[t]), want, topk)
                assert [(g.Id, g.Score) for g in one] == [(g.Id, g.Score) for g in got]


# ---- logics/item_to_item_test.go ------------------------------------------------------------------------------------
def item_to_item_column(db):  # TestColumnFunc (item_to_item_test.go:46-115): what reaches the vector database
    i2i = V.EmbeddingItemToItem("column", 1_790_000_000_000, db)
    coll = V.ItemToItemCollection("column")
    i2i.Add("1", [0.1, 0.2, 0.3])
    i2i.Clean()
    assert db.CountVectors(coll) == 1
    i2i.Add("2", [0.1, 0.2, 0.3], is_hidden=True)  # hidden items are stored (and never returned)
    i2i.Clean()
    assert db.CountVectors(coll) == 2
    i2i.Add("1", [0.1, 0.2])  # dimension does not match: dropped
    i2i.Clean()
    assert db.CountVectors(coll) == 2
    i2i.Add("1", "hello")  # type does not match
    i2i.Clean()
    assert db.CountVectors(coll) == 2
    i2i.Add("2", None)  # column does not exist
    i2i.Clean()
    assert db.CountVectors(coll) == 2


def item_to_item_embedding(db):  # TestEmbedding (item_to_item_test.go:117-141)
    i2i = V.EmbeddingItemToItem("embedding", 1_790_000_000_000, db)
    for i in range(100):
        f = np.float32(i)
        i2i.Add(str(i), [np.float32(0.1) * f, np.float32(0.2) * f, np.float32(0.3) * f])
    i2i.Clean()
    coll = V.ItemToItemCollection("embedding")
    scores = V.QuerySimilar(db, coll, "0", None, 10)
    assert [s.Id for s in scores] == [str(i) for i in range(1, 11)]
    assert all(0 < s.Score < 1 for s in scores) and scores[0].Score > scores[-1].Score  # 1 / (1 + distance)
    assert V.QuerySimilar(db, coll, "no-such-item", None, 10) == []
    bulk = V.QuerySimilarBulk(db, coll, [str(i) for i in range(100)] + ["missing"], None, 5)
    assert bulk[-1] == [] and len(bulk) == 101
    for i in (0, 1, 50, 99):
        one = V.QuerySimilar(db, coll, str(i), None, 5)
        assert [(s.Id, s.Score) for s in bulk[i]] == [(s.Id, s.Score) for s in one]
    assert [s.Id for s in bulk[50]][:2] in (["49", "51"], ["51", "49"])


def item_to_item_clean(db):  # TestClean (item_to_item_test.go:143-170): Clean drops what an earlier refresh left behind
    ts = 1_790_000_000_000
    coll = V.ItemToItemCollection("cleanup")
    db.AddCollection(coll, 2, V.Euclidean)
    db.AddVectors(coll, [V.Vector("stale", [0, 0], Timestamp=ts - 3_600_000), V.Vector("fresh", [1, 1], Timestamp=ts)])
    i2i = V.EmbeddingItemToItem("cleanup", ts, db)
    i2i.Clean()
    assert db.CountVectors(coll) == 1 and db.GetVectors(coll, ["stale", "fresh"])[0].Id == "fresh"
    # a collection with another dimension is recreated by the writer (vector_writer.go:127-137)
    i2i3 = V.EmbeddingItemToItem("cleanup", ts + 1, db)
    i2i3.Add("x", [1, 2, 3])
    i2i3.Clean()
    assert db.DescribeCollection(coll)["Dimension"] == 3 and db.CountVectors(coll) == 1


# ---- logics/user_to_user_test.go (the embedding kind is the same writer and query over the user_to_user_* collection) --
def user_to_user_embedding(db):  # TestEmbedding (user_to_user_test.go:48-71)
    coll = V.UserToUserCollection("embedding")
    u2u = V.EmbeddingItemToItem("embedding", 1_790_000_000_000, db, collection=coll)
    for i in range(100):
        f = np.float32(i)
        u2u.Add(str(i), [np.float32(0.1) * f, np.float32(0.2) * f, np.float32(0.3) * f])
    u2u.Clean()
    scores = V.QuerySimilar(db, coll, "0", None, 10)
    assert [s.Id for s in scores] == [str(i) for i in range(1, 11)]


def user_to_user_clean(db):  # TestClean (user_to_user_test.go:73-93)
    ts = 1_790_000_000_000
    coll = V.UserToUserCollection("cleanup")
    db.AddCollection(coll, 2, V.Euclidean)
    db.AddVectors(coll, [V.Vector("stale", [0, 0], Timestamp=ts - 3_600_000), V.Vector("current", [1, 0], Timestamp=ts)])
    w = V.EmbeddingItemToItem("cleanup", ts, db, collection=coll)
    w.Add("new", [2, 0])
    w.Clean()
    assert [v.Id for v in db.GetVectors(coll, ["stale", "current", "new"])] == ["current", "new"]


# ---- master/tasks.go:930-962 + worker/pipeline.go:403-425: item factors in a Dot collection, per-user recommendations ----
def collaborative_recommend(db):
    rng = np.random.default_rng(12)
    n_items, n_users, d, cache_size = 400, 90, 16, 10
    Qf = (rng.standard_normal((n_items, d)) * 0.3).astype(np.float32)
    P = (rng.standard_normal((n_users, d)) * 0.3).astype(np.float32)
    coll = V.CollaborativeFilteringCollection(1790000000000)
    db.AddCollection(coll, d, V.Dot)
    for start in range(0, n_items, 128):  # batches, as the master's index loop
        db.AddVectors(coll, [V.Vector("i%d" % i, Qf[i], IsHidden=(i % 17 == 0), Categories=["c%d" % (i % 3)])
                             for i in range(start, min(start + 128, n_items))])
    excludes = [["i%d" % i for i in rng.choice(n_items, int(rng.integers(0, 25)), replace=False)] for _ in range(n_users)]
    bulk = V.CollaborativeRecommendBulk(db, coll, P, excludes, cache_size)
    assert len(bulk) == n_users
    for u in range(n_users):
        # the reference's per-user form: QueryVectors(CacheSize + |exclude|), then drop the excluded ids
        ref = [v for v in db.QueryVectors(coll, V.Vector(Values=P[u]), None, cache_size + len(excludes[u])) if v.Id not in excludes[u]]
        assert [s.Id for s in bulk[u]] == [v.Id for v in ref], u
        assert [np.float32(s.Score) for s in bulk[u]] == [np.float32(v.Score) for v in ref]
        assert len(bulk[u]) >= cache_size and not any(s.Id in excludes[u] for s in bulk[u])
        assert all(int(s.Id[1:]) % 17 != 0 for s in bulk[u])  # hidden items never recommended
        # scores are the inner products (Dot), best first
        want = sorted((float(np.float32(np.dot(P[u].astype(np.float64), Qf[int(s.Id[1:])].astype(np.float64)))) for s in bulk[u]), reverse=True)
        assert np.allclose([s.Score for s in bulk[u]], want, rtol=1e-5, atol=1e-6)
