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


def sparse(db):  # TestSparse (database_test.go:187-236)
    db.AddCollection("test_sparse", 0, V.Dot)
    info = db.DescribeCollection("test_sparse")
    assert info["Dimension"] == 0 and info["Distance"] == V.Dot
    cutoff = 1_790_000_000_000
    db.AddVectors("test_sparse", [V.Vector("old", [1, 1], Indices=[1, 100], Timestamp=cutoff - 3_600_000),
                                  V.Vector("match", [1, 2], Indices=[1, 100], Timestamp=cutoff),
                                  V.Vector("other", [1, 2], Indices=[2, 200], Timestamp=cutoff)])
    assert db.CountVectors("test_sparse") == 3
    got = db.GetVectors("test_sparse", ["match", "missing"])
    assert len(got) == 1 and got[0].Id == "match" and got[0].Indices == [1, 100] and got[0].Values == [1, 2]
    assert got[0].Timestamp == cutoff
    r = db.QueryVectors("test_sparse", V.Vector(Values=[1, 2], Indices=[1, 100]), None, 10)
    assert len(r) == 2 and r[0].Id == "match"  # "other" shares no index: Score == 0 is dropped (xvec.go:419-421)
    assert [x.Id for x in r] == ["match", "old"] and [x.Score for x in r] == [5.0, 3.0] and r[0].Indices == [1, 100]
    db.DeleteVectors("test_sparse", cutoff)
    assert db.CountVectors("test_sparse") == 2
    db.DeleteCollection("test_sparse")
    with pytest.raises(V.ErrNotFound):
        db.DescribeCollection("test_sparse")


def sparse_rules(db):  # xvec.go:241-247 (sparse = Dot only), :301-327 (validation before the upsert), :386-394 (filters)
    with pytest.raises(V.ErrNotSupported):
        db.AddCollection("s", 0, V.Cosine)  # "distance method for sparse vector"
    with pytest.raises(V.ErrNotSupported):  # quantized collections (database_test.go:332-420)
        db.AddCollection("q", defaultVectorSize, V.Cosine, quantization="sq", bits=8)
    db.AddCollection("s", 0, V.Dot)
    db.AddCollection("dense", defaultVectorSize, V.Dot)
    with pytest.raises(ValueError):
        db.AddVectors("dense", [V.Vector("x", [1, 1], Indices=[1, 100])])  # a sparse vector in a dense collection
    with pytest.raises(ValueError):
        db.AddVectors("s", [V.Vector("ok", [1.0], Indices=[3]), V.Vector("x", [1, 0, 0, 0])])  # dense in sparse: nothing added
    with pytest.raises(ValueError):
        db.AddVectors("s", [V.Vector("x", [1, 2], Indices=[7, 7])])  # repeated index
    assert db.CountVectors("s") == 0
    with pytest.raises(ValueError):
        db.QueryVectors("dense", V.Vector(Values=[1, 2], Indices=[1, 100]), None, 10)
    # entries may come in any order; hidden vectors and the categories filter; upsert; zero / negative scores
    db.AddVectors("s", [V.Vector("a", [2, 1], Indices=[9, 4], Categories=["c", "x"]),
                        V.Vector("b", [1, 1], Indices=[4, 9], Categories=["c"]),
                        V.Vector("h", [5, 5], Indices=[4, 9], IsHidden=True, Categories=["c", "x"]),
                        V.Vector("n", [-1], Indices=[4], Categories=["c"]),
                        V.Vector("z", [1, -1], Indices=[4, 9], Categories=["c"]),
                        V.Vector("far", [1], Indices=[1000])])
    q = V.Vector(Values=[1, 1], Indices=[9, 4])
    assert [(x.Id, x.Score) for x in db.QueryVectors("s", q, None, 10)] == [("a", 3.0), ("b", 2.0), ("n", -1.0)]
    # the reference ranks all five visible vectors, cuts, THEN drops the zero scores ("z" cancels, "far" is disjoint):
    # with topK 4 the slots after "a", "b" go to the two zeros and "n" is not returned
    assert [x.Id for x in db.QueryVectors("s", q, None, 4)] == ["a", "b"]
    assert [x.Id for x in db.QueryVectors("s", q, ["x"], 10)] == ["a"]
    assert db.QueryVectors("s", q, ["nope"], 10) == [] and db.QueryVectors("s", q, None, 0) == []
    db.AddVectors("s", [V.Vector("b", [4], Indices=[9], Categories=["c"])])  # upsert
    assert db.CountVectors("s") == 6
    assert [(x.Id, x.Score) for x in db.QueryVectors("s", q, None, 2)] == [("b", 4.0), ("a", 3.0)]
    bulk = db.QuerySparseBatch("s", [q, V.Vector(Values=[1], Indices=[1000]), V.Vector(Values=[1], Indices=[77])], None, 3)
    assert [[x.Id for x in r] for r in bulk] == [["b", "a"], ["far"], []]


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


def item_to_item_hidden(db):  # TestHidden (item_to_item_test.go:166-210)
    i2i = V.EmbeddingItemToItem("hidden", 1_790_000_000_000, db)
    i2i.Add("visible_1", [0.0, 0.0, 0.0])
    i2i.Add("visible_2", [0.1, 0.0, 0.0])
    i2i.Add("hidden_1", [0.05, 0.0, 0.0], is_hidden=True)
    i2i.Clean()
    coll = V.ItemToItemCollection("hidden")
    hidden_scores = V.QuerySimilar(db, coll, "hidden_1", None, 2)  # a hidden item still gets neighbours, from the visible ones
    assert len(hidden_scores) == 2 and all(s.Id != "hidden_1" for s in hidden_scores)
    visible_scores = V.QuerySimilar(db, coll, "visible_1", None, 2)  # and is never anybody's neighbour
    assert [s.Id for s in visible_scores] == ["visible_2"]


def _nested_kind(db, kind, collection, n=100):
    """TestTags / TestUsers (item_to_item_test.go:212-270) and TestTags / TestItems (user_to_user_test.go:96-153): entity i
    carries the ids 1 .. 100-i (as labels or as feedback), every idf is 1; the neighbours of "0" are "1" .. "10" in order"""
    idf = np.ones(101, np.float32)
    w = V.SparseSimilarity(kind, collection, 1_790_000_000_000, db, tags_idf=idf if kind == "tags" else None,
                           feedback_idf=None if kind == "tags" else idf)
    for i in range(n):
        ids = list(range(100 - i, 0, -1))  # unsorted on purpose: the writer sorts (slices.Sort)
        w.Add(str(i), tags=ids if kind == "tags" else (), feedback=() if kind == "tags" else ids)
    w.Clean()
    scores = V.QuerySimilarTyped(db, collection, kind, "0", None, 10)
    assert [s.Id for s in scores] == [str(i) for i in range(1, 11)]
    assert [s.Score for s in scores] == [float(100 - i) for i in range(1, 11)]  # sum of idf over the common ids
    bulk = V.QuerySimilarTypedBulk(db, collection, kind, [str(i) for i in range(n)] + ["missing"], None, 7)
    assert bulk[-1] == [] and len(bulk) == n + 1
    for i in (0, 1, 37, 99):
        one = V.QuerySimilarTyped(db, collection, kind, str(i), None, 7)
        assert [(s.Id, s.Score) for s in bulk[i]] == [(s.Id, s.Score) for s in one]
    assert V.QuerySimilarTyped(db, collection, kind, "no-such-id", None, 10) == []


def item_to_item_tags(db):  # TestTags (item_to_item_test.go:212-242)
    _nested_kind(db, "tags", V.ItemToItemCollection("tags"))


def item_to_item_users(db):  # TestUsers (item_to_item_test.go:244-271)
    _nested_kind(db, "users", V.ItemToItemCollection("users"))


def _auto_kind(db, collection):
    """TestAuto (item_to_item_test.go:273-316): even entities carry labels, odd ones feedback; both id spaces share one
    vector (feedback ids offset by len(tagsIDF)), so even entities only match even ones, odd only odd; scores halved"""
    idf = np.ones(101, np.float32)
    w = V.SparseSimilarity("auto", collection, 1_790_000_000_000, db, tags_idf=idf, feedback_idf=idf)
    for i in range(100):
        ids = list(range(1, 100 - i + 1))
        w.Add(str(i), tags=ids if i % 2 == 0 else (), feedback=() if i % 2 == 0 else ids)
    w.Clean()
    s0 = V.QuerySimilarTyped(db, collection, "auto", "0", None, 10)
    assert [s.Id for s in s0] == [str(2 * i) for i in range(1, 11)]
    assert [s.Score for s in s0] == [(100 - 2 * i) * .5 for i in range(1, 11)]
    s1 = V.QuerySimilarTyped(db, collection, "auto", "1", None, 10)
    assert [s.Id for s in s1] == [str(2 * i + 1) for i in range(1, 11)]
    v = db.GetVectors(collection, ["1"])[0]
    assert v.Indices[0] == 101 + 1 and len(v.Indices) == 99  # offset = len(tagsIDF)


def item_to_item_auto(db):
    _auto_kind(db, V.ItemToItemCollection("auto"))


def item_to_item_sparse_hidden_and_idf(db):
    """TestHidden's rule for a sparse kind (item_to_item_test.go:166-210: hidden items are stored, never returned), ids with
    idf <= 0 or outside the table dropped (vector_writer.go:200-208), an item without usable ids not written (:88-91),
    and the categories filter of QueryItemToItem"""
    coll = V.ItemToItemCollection("users_idf")
    idf = np.array([0.0, 4.0, 9.0, -1.0, 16.0], np.float32)  # users 0 and 3 carry no weight
    w = V.SparseSimilarity("users", coll, 1_790_000_000_000, db, feedback_idf=idf)
    w.Add("a", feedback=[1, 2, 4], categories=["k"])
    w.Add("b", feedback=[4, 0, 1, 99], categories=["k"])      # 0: idf 0, 99: outside the table
    w.Add("c", feedback=[2, 3], is_hidden=True, categories=["k"])
    w.Add("d", feedback=[2])
    w.Add("e", feedback=[0, 3, 7])                               # nothing usable: no vector
    w.Clean()
    assert db.CountVectors(coll) == 4
    vb = db.GetVectors(coll, ["b"])[0]
    assert vb.Indices == [1, 4] and vb.Values == [2.0, 4.0]  # sqrt(idf)
    r = V.QuerySimilarTyped(db, coll, "users", "a", None, 10)
    assert [(s.Id, s.Score) for s in r] == [("b", 20.0), ("d", 9.0)]  # 4 + 16, 9; "c" is hidden; "a" itself skipped
    assert [s.Id for s in V.QuerySimilarTyped(db, coll, "users", "a", ["k"], 10)] == ["b"]
    assert V.QuerySimilarTyped(db, coll, "users", "e", None, 10) == []


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


def user_to_user_tags(db):  # TestTags (user_to_user_test.go:96-125)
    _nested_kind(db, "tags", V.UserToUserCollection("tags"))


def user_to_user_items(db):  # TestItems (user_to_user_test.go:127-153)
    _nested_kind(db, "items", V.UserToUserCollection("items"))


def user_to_user_auto(db):  # TestAuto (user_to_user_test.go:155-190)
    _auto_kind(db, V.UserToUserCollection("auto"))


def publish_and_recommend(db):
    """master/tasks.go:925-969 then worker/pipeline.go:403-425: a fitted model is published (items -> Dot collection, users ->
    blob), a worker restores the blob and asks for every user's recommendations in one bulk search"""
    from gorse_amd import cf
    rng = np.random.default_rng(21)
    U, I, d = 40, 150, 8
    P = (rng.standard_normal((U, d)) * 0.3).astype(np.float32)
    Q = (rng.standard_normal((I, d)) * 0.3).astype(np.float32)
    m = cf.BPR({"NFactors": d})
    m.load_factors(P, Q)  # ids "0".."n-1", every row predictable
    hidden = [i % 11 == 0 for i in range(I)]
    cats = [["c%d" % (i % 4)] for i in range(I)]
    model_id = 1_790_000_123_456
    users = V.PublishCollaborativeFiltering(m, db, model_id, hidden, cats, batch_size=64)
    coll = V.CollaborativeFilteringCollection(model_id)
    assert db.CountVectors(coll) == I and users.Count() == U
    info = db.DescribeCollection(coll)
    assert info["Dimension"] == d and info["Distance"] == V.Dot
    v7 = db.GetVectors(coll, ["7", "11"])
    assert v7[0].Values == [float(x) for x in Q[7]] and v7[0].Categories == ["c3"] and not v7[0].IsHidden and v7[1].IsHidden
    assert v7[0].Timestamp == model_id
    worker = V.MatrixFactorizationUsers()
    worker.Unmarshal(users.Marshal())
    emb = np.stack([worker.Get(str(u))[0] for u in range(U)])
    assert np.array_equal(emb, P) and worker.Get("nobody") == (None, False)
    rec = V.CollaborativeRecommendBulk(db, coll, emb, [[] for _ in range(U)], 5)
    for u in range(U):
        scores = Q.astype(np.float64) @ P[u].astype(np.float64)
        best = [i for i in np.argsort(-scores, kind="stable") if not hidden[i]][:5]
        assert [s.Id for s in rec[u]] == [str(i) for i in best], u


# dense cases written after the last device session: on the CPU list now, on the device through tests/test_gpu_vectors_sparse.py
PENDING_DENSE_CASES = [item_to_item_hidden, publish_and_recommend]
SPARSE_CASES = [sparse, sparse_rules, item_to_item_tags, item_to_item_users, item_to_item_auto,
                item_to_item_sparse_hidden_and_idf, user_to_user_tags, user_to_user_items, user_to_user_auto]


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
