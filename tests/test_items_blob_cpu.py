"""logics.MatrixFactorizationItems blob (gorse_amd/host/gorse_vectors.hpp): a file written by a build WITHOUT this library --
the reference's stream, logics/cf.go:81-101 with the HNSW graph of common/ann/hnsw.go:278-337 inside -- is read here (vectors
and ids kept, the graph skipped), and the file this library writes keeps the reference's framing around an index section whose first "vector"
is an empty gob stream: the reference's reader returns gob's EOF there, an ordinary error.  The reference stream is built below byte by byte from
encoding/gob's documented wire format, independently of host/gob.hpp (no Go toolchain here: unpinned against a Go encoder)."""
import struct

import numpy as np
import pytest

from gorse_amd import vectors as V
from oracle import oracle as orc


# ---- encoding/gob, restated from the package documentation ("Encoding Details") ----
def g_uint(v):
    if v < 128:
        return bytes([v])
    b = v.to_bytes((v.bit_length() + 7) // 8, "big")
    return bytes([256 - len(b)]) + b


def g_int(i):
    return g_uint((~i << 1) | 1 if i < 0 else i << 1)


def g_float(f):
    return g_uint(int.from_bytes(struct.pack("<d", f), "big"))  # float64 bits byte-reversed


def g_msg(body):
    return g_uint(len(body)) + body


def gob_int(v):
    return g_msg(g_int(2) + b"\x00" + g_int(v))


def gob_string(s):
    return g_msg(g_int(6) + b"\x00" + g_uint(len(s.encode())) + s.encode())


def gob_f32_slice(v):
    common = g_uint(1) + g_uint(len(b"[]float32")) + b"[]float32" + g_uint(1) + g_int(65) + g_uint(0)
    slice_type = g_uint(1) + common + g_uint(1) + g_int(4) + g_uint(0)  # CommonType, Elem = float
    definition = g_int(-65) + g_uint(2) + slice_type + g_uint(0)  # wireType.SliceT is field 1
    value = g_int(65) + b"\x00" + g_uint(len(v)) + b"".join(g_float(float(np.float32(x))) for x in v)
    return g_msg(definition) + g_msg(value)


def gob_time(unix_sec, nanos):
    # a GobEncoder value: wireType{GobEncoderT: gobEncoderType{CommonType{Name: "Time", Id: 65}}} (field 4), then type id 65,
    # the singleton marker, the byte count and time.Time.MarshalBinary version 1 (UTC: zone offset -1)
    common = g_uint(1) + g_uint(4) + b"Time" + g_uint(1) + g_int(65) + g_uint(0)
    definition = g_int(-65) + g_uint(5) + g_uint(1) + common + g_uint(0) + g_uint(0)
    payload = bytes([1]) + (unix_sec + 62135596800).to_bytes(8, "big") + nanos.to_bytes(4, "big") + (0xFFFF).to_bytes(2, "big")
    return g_msg(definition) + g_msg(g_int(65) + b"\x00" + g_uint(len(payload)) + payload)


def write_gob(blob):  # encoding.WriteGob: int32 byte count + the stream
    return struct.pack("<i", len(blob)) + blob


def pq(elems, desc=False):  # PriorityQueue.Marshal, common/heap/pq.go:128-133
    return struct.pack("<?i", desc, len(elems)) + b"".join(struct.pack("<if", v, w) for v, w in elems)


def reference_stream(ts, X, ids, rng):
    """logics/cf.go:81-101 around hnsw.go:278-337, with a plausible (random) graph"""
    n, d = X.shape
    out = write_gob(gob_time(*ts)) + write_gob(gob_int(d))
    out += struct.pack("<fqqqq", 1.0 / np.log(48.0), 16, 32, 0, 100) + struct.pack("<q", n)
    for row in X:
        out += write_gob(gob_f32_slice(row))
    for i in range(n):
        nb = rng.choice(n, int(rng.integers(0, min(n, 9))), replace=False)
        out += pq([(int(j), float(rng.random())) for j in nb])
    layers = 2
    out += struct.pack("<q", layers)
    for l in range(layers):
        members = rng.choice(n, max(1, n // (4 ** (l + 1))), replace=False)
        out += struct.pack("<i", members.size)
        for m in members:
            out += struct.pack("<i", int(m)) + pq([(int(j), 0.5) for j in members[:3]])
    out += struct.pack("<i", 0)  # enterPoint
    out += write_gob(gob_int(n))
    for s in ids:
        out += write_gob(gob_string(s))
    return out


def reference_reader_outcome(blob):
    """what the reference's MatrixFactorizationItems.Unmarshal (logics/cf.go:103-128, hnsw.go:338-418) does with a file, as far as
    its first failure: every ReadGob of a non-empty stream is taken as decodable"""
    at = 0

    def read_gob():
        nonlocal at
        (ln,) = struct.unpack_from("<i", blob, at)
        at += 4
        if ln == 0:
            return False  # gob.NewDecoder(empty).Decode -> io.EOF
        at += ln
        return True
    for what in ("timestamp", "dimension"):
        if not read_gob():
            return "error at %s: gob EOF" % what
    at += 4 + 4 * 8
    (nv,) = struct.unpack_from("<q", blob, at)
    at += 8
    for i in range(nv):
        if not read_gob():
            return "error at vector %d: gob EOF" % i
    for i in range(nv):
        (ln,) = struct.unpack_from("<i", blob, at + 1)
        at += 5 + 8 * ln
    (layers,) = struct.unpack_from("<q", blob, at)
    at += 8
    for _ in range(layers):
        (m,) = struct.unpack_from("<i", blob, at)
        at += 4
        for _ in range(m):
            (ln,) = struct.unpack_from("<i", blob, at + 5)
            at += 4 + 5 + 8 * ln
    at += 4
    read_gob()
    while at < len(blob):
        read_gob()
    return "ok" if at == len(blob) else "trailing bytes"


def cpu_searcher(oracle):
    def searcher(X, n, d, metric, Q, nq, k, idx, dist, cnt):
        Xa = np.ctypeslib.as_array(X, (n, d)).copy()
        Qa = np.ctypeslib.as_array(Q, (nq, d)).copy()
        I, D, Cn = np.ctypeslib.as_array(idx, (nq, k)), np.ctypeslib.as_array(dist, (nq, k)), np.ctypeslib.as_array(cnt, (nq,))
        for t in range(nq):
            ei, ed = oracle.search_vector(Xa, metric, Qa[t], k)
            I[t, :ei.size], D[t, :ei.size], Cn[t] = ei, ed, ei.size
        return 0
    return searcher


def test_reads_the_reference_stream_and_writes_a_framing_the_reference_rejects(oracle):
    oracle.set_isa(orc.ISA_AVX512)
    rng = np.random.default_rng(12)
    n, d = 57, 16
    X = rng.standard_normal((n, d)).astype(np.float32)
    X[3, 0], X[4, 1] = 0.0, -1e-30  # a zero and a tiny value: the shortest and a long gob float
    ids = ["item-%d" % i for i in range(n)]
    ids[7] = "ünïcode / id"
    ts = (1_790_000_000, 123_456_789)
    blob = reference_stream(ts, X, ids, rng)
    m = V.MatrixFactorizationItems(searcher=cpu_searcher(oracle))
    m.Unmarshal(blob)
    assert m.Count() == n and m.Dimension() == d
    assert m.Timestamp() == ts[0] * 1_000_000_000 + ts[1]
    for i in range(n):
        assert m.Id(i) == ids[i]
        assert np.array_equal(m.Row(i).view(np.uint32), X[i].view(np.uint32)), i
    # Search = the exact nearest items by -dot, Score = the inner product (cf.go:69-79)
    q = rng.standard_normal(d).astype(np.float32)
    got = m.Search(q, 5)
    ei, ed = oracle.search_vector(X, orc.METRIC_NEG_DOT, q, 5)
    assert [g[0] for g in got] == [ids[i] for i in ei]
    assert np.array_equal(np.array([g[1] for g in got]), -ed.astype(np.float64))
    # the file this library writes: the reference's framing around an index section a reference reader rejects
    out = m.Marshal()
    assert reference_reader_outcome(out) == "error at vector 0: gob EOF"
    assert reference_reader_outcome(blob) == "ok"
    head = len(write_gob(gob_time(*ts))) + len(write_gob(gob_int(d)))
    assert out[:head] == blob[:head]  # timestamp and dimension exactly as the reference writes them
    assert out[head:head + 4] == b"GHIP" and struct.unpack_from("<q", out, head + 4)[0] == 1
    tail = b"".join(write_gob(gob_string(s_)) for s_ in ids)
    assert out.endswith(write_gob(gob_int(n)) + tail) and blob.endswith(write_gob(gob_int(n)) + tail)
    m2 = V.MatrixFactorizationItems(searcher=cpu_searcher(oracle))
    m2.Unmarshal(out)
    assert m2.Count() == n and m2.Timestamp() == m.Timestamp()
    assert all(m2.Id(i) == ids[i] and np.array_equal(m2.Row(i), X[i]) for i in range(n))
    assert m2.Search(q, 5) == got
    bad = bytearray(out)
    bad[head + 4] = 9  # an index section of a later version
    with pytest.raises(Exception):
        V.MatrixFactorizationItems().Unmarshal(bytes(bad))
    with pytest.raises(Exception):
        V.MatrixFactorizationItems().Unmarshal(blob[:len(blob) // 2])  # a truncated reference stream


def test_add_search_marshal_from_scratch(oracle):
    oracle.set_isa(orc.ISA_AVX512)
    rng = np.random.default_rng(5)
    m = V.MatrixFactorizationItems(timestamp_unix_nanos=42, searcher=cpu_searcher(oracle))
    X = rng.standard_normal((20, 8)).astype(np.float32)
    for i, row in enumerate(X):
        m.Add("i%d" % i, row)
    m.Add("wrong-dimension", np.zeros(3, np.float32))  # logged and dropped (cf.go:56-61)
    assert m.Count() == 20
    got = m.Search(X[4], 3)
    ei, ed = oracle.search_vector(X, orc.METRIC_NEG_DOT, X[4], 3)
    assert [g[0] for g in got] == ["i%d" % i for i in ei]
    m2 = V.MatrixFactorizationItems(searcher=cpu_searcher(oracle))
    m2.Unmarshal(m.Marshal())
    assert m2.Timestamp() == 42 and m2.Search(X[4], 3) == got


def _index_section_start(blob):
    at = 0
    for _ in range(2):  # timestamp, dimension
        (ln,) = struct.unpack_from("<i", blob, at)
        at += 4 + ln
    return at


def test_marshal_reference_writes_a_graph_the_reference_can_walk(oracle):
    """MarshalReference: the blob in the reference's own format (logics/cf.go:81-101 around HNSW.Marshal, hnsw.go:278-337) with a
    graph built from exact all-pairs searches -- what a master with this library hands to workers WITHOUT it.  Checked here: the
    reference's reader walks the whole file; NewHNSW's parameters; every queue is a valid heap array (ascending distances), holds
    no self loop and at most maxConnection0 / maxConnection entries whose weights are the -dot distances; the layers nest; and the
    reference's own knnSearch (restated: oracle.hnsw_knn_search) finds the exact nearest items on it.  This library's reader
    takes the file like any reference file (vectors and ids kept)."""
    oracle.set_isa(orc.ISA_AVX512)
    rng = np.random.default_rng(21)
    n, d = 700, 12
    X = rng.standard_normal((n, d)).astype(np.float32)
    ids = ["item-%d" % i for i in range(n)]
    m = V.MatrixFactorizationItems(timestamp_unix_nanos=1_790_000_000_123_456_789, searcher=cpu_searcher(oracle))
    for i in range(n):
        m.Add(ids[i], X[i])
    blob = m.MarshalReference()
    assert reference_reader_outcome(blob) == "ok"
    params, streams, bottom, upper, enter, end = orc.hnsw_parse_index_section(blob, _index_section_start(blob))
    assert params["maxConnection"] == 48 and params["maxConnection0"] == 96 and params["efConstruction"] == 100 and params["ef"] == 0
    assert abs(params["levelFactor"] - 1.0 / np.log(48.0)) < 1e-6
    assert len(streams) == n and streams[5] == gob_f32_slice(X[5])  # the vectors as the reference writes them
    indeg = np.zeros(n, np.int64)
    for i, (desc, q) in enumerate(bottom):
        # a queue = the 84 nearest others (exact) + at most 12 reverse links, ascending, every weight the pair's own -dot distance
        assert not desc and 84 <= len(q) <= 96 and all(v != i and 0 <= v < n for v, _ in q) and len({v for v, _ in q}) == len(q)
        w = [x for _, x in q]
        assert w == sorted(w)
        ei, ed = oracle.search_vector(X, orc.METRIC_NEG_DOT, X[i], 85)
        want = [(int(j), np.float32(dd)) for j, dd in zip(ei, ed) if j != i][:84]
        have = {v: np.float32(x) for v, x in q}
        assert all(have.get(j) == dd for j, dd in want), i
        exact = dict(want)
        for v, x in q:
            if v not in exact:  # a reverse link: i is among v's nearest, and the weight is the same distance
                assert np.float32(x) == np.float32(-np.dot(X[i].astype(np.float64), X[v].astype(np.float64))) or abs(x + float(X[i] @ X[v])) < 1e-5
            indeg[v] += 1
    assert (indeg > 0).all()  # every vector can be reached
    assert len(upper) >= 1 and enter in upper[-1]
    for L, layer in enumerate(upper):
        members = set(layer)
        if L > 0:
            assert members <= set(upper[L - 1])  # a vector of level >= L + 1 is also in layer L
        for key, (desc, q) in layer.items():
            assert not desc and len(q) <= 48 and all(v in members and v != key for v, _ in q)
            assert [x for _, x in q] == sorted(x for _, x in q)
    # the reference's search on this graph: the exact answer (nearly every pair is linked at this size)
    for t in range(40):
        qv = rng.standard_normal(d).astype(np.float32)
        got = orc.hnsw_knn_search(X, bottom, upper, enter, params, qv, 5)
        ei, _ = oracle.search_vector(X, orc.METRIC_NEG_DOT, qv, 5)
        assert got == [int(j) for j in ei], t
    m2 = V.MatrixFactorizationItems(searcher=cpu_searcher(oracle))
    m2.Unmarshal(blob)
    assert m2.Count() == n and m2.Timestamp() == m.Timestamp()
    assert all(m2.Id(i) == ids[i] and np.array_equal(m2.Row(i), X[i]) for i in range(0, n, 37))
    # nothing stored: an empty graph the reference reads as well
    e = V.MatrixFactorizationItems(searcher=cpu_searcher(oracle))
    assert reference_reader_outcome(e.MarshalReference()) == "ok"


def test_reference_levels_follow_the_arithmetic_the_go_twin_writes():
    """The level stream of MarshalReference (hnsw.go:137 on a splitmix64 stream): the C++ twin's draws equal a numpy restatement of the
    Go twin's expression -- int(math.Floor(float64(-float32(math.Log(float64(u))) * levelFactor))), levelFactor = float32(1 /
    math.Log(48)) -- for a million vectors, including the draws whose product lies within a few ulps of an integer (where a float
    logf on one side would flip a level)."""
    import ctypes as C

    from gorse_amd import cf
    H = cf.host()
    H.gh_test_hnsw_levels.restype = C.c_uint32
    H.gh_test_hnsw_levels.argtypes = [C.c_int64, C.c_void_p]
    for n in (1, 1000, 1_000_000):
        got = np.empty(n, np.int32)
        bits = H.gh_test_hnsw_levels(n, got.ctypes.data)
        factor = np.float32(1.0 / np.log(np.float64(48.0)))
        assert bits == int(factor.view(np.uint32))
        M = (1 << 64) - 1
        st = (0x9E3779B97F4A7C15 ^ n) & M
        z = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(st))  # wraps modulo 2^64
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
        u = ((z >> np.uint64(40)).astype(np.float32) + np.float32(1.0)) / np.float32(16777216.0)
        lg = np.log(u.astype(np.float64)).astype(np.float32)
        want = np.floor((-lg * factor).astype(np.float64)).astype(np.int32)
        assert np.array_equal(got, want)
        if n == 1_000_000:
            prod = (-lg * factor).astype(np.float64)
            near = np.abs(prod - np.round(prod)) < 1e-5  # the draws a one-ulp difference of the logarithm could flip
            print("levels of %d vectors: top %d, %d draws within 1e-5 of an integer" % (n, got.max(), int(near.sum())))
            assert got.max() >= 3 and (got == 0).mean() > 0.97
