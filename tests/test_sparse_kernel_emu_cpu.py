"""CPU check of the sparse top-k KERNEL's control flow (gorse_amd/csrc/sparse_kernels.hpp) without a GPU: the kernel
source is compiled for the host through tests/emu/hip_emu.hpp (one OS thread per work-item, spin barriers) together
with the product's own index builder (sparse_host.hpp) and compared with the oracle bit for bit.  This exercises
barrier placement, the slot hand-out / overflow / cut logic of the ranking buffer, the stamp / touched bookkeeping
across queries and launches, padding and the statistics; it says nothing about gfx950 code generation or speed --
tests/test_gpu_vectors_sparse.py is the parity test proper."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from sparse_cases import TIE_EXPECT, check_against_oracle as check, random_csr, rows_of, tie_case

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(EMU, "libsparse_emu.so")
    srcs = [os.path.join(EMU, "sparse_emu.cpp"), os.path.join(EMU, "hip_emu.hpp"),
            os.path.join(HERE, "..", "gorse_amd", "csrc", "sparse_kernels.hpp"),
            os.path.join(HERE, "..", "gorse_amd", "csrc", "sparse_host.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-pthread", "-o", so,
                               srcs[0]])
    L = C.CDLL(so)
    p = C.c_void_p
    L.emu_sparse_search.restype = C.c_int
    L.emu_sparse_search.argtypes = [C.c_int64, p, p, p, C.c_int64, p, p, p, C.c_int64, p, C.c_int, p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_uint32, p, p, p, p]
    return L


def run_emu(L, ptr, idx, val, k, nq, q=None, q_first=0, exclude=None, exclude_self=0, mask=None, grid=3, block=8,
            rounds=1, serial_base=0):
    N = ptr.size - 1
    out_idx = np.full((nq, k), -7, dtype=np.int32)
    out_sc = np.zeros((nq, k), dtype=np.float32)
    out_cnt = np.full(nq, -7, dtype=np.int32)
    stat = np.zeros(2, dtype=np.uint64)
    ptr_ = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    qp, qi, qv = q if q is not None else (None, None, None)
    rc = L.emu_sparse_search(N, ptr_(ptr), ptr_(idx), ptr_(val), nq, ptr_(qp), ptr_(qi), ptr_(qv), q_first,
                             ptr_(exclude), exclude_self, ptr_(mask), k, grid, block, rounds, serial_base, ptr_(out_idx),
                             ptr_(out_sc), ptr_(out_cnt), ptr_(stat))
    assert rc == 0
    return out_idx, out_sc, out_cnt, stat


@pytest.mark.parametrize("block", [1, 5, 8, 64])
def test_all_pairs_equals_oracle(emu, oracle, block):
    rng = np.random.default_rng(100 + block)
    ptr, idx, val = random_csr(rng, 70, 40, 0, 9)
    k = 7
    got = run_emu(emu, ptr, idx, val, k, 70, exclude_self=1, grid=4, block=block)
    queries = rows_of(ptr, idx, val, range(70))
    check(oracle, ptr, idx, val, k, got, queries, list(range(70)), None)
    # statistics: multiply-adds = sum of posting-list lengths over the queries' indices; hits incl. the row itself
    lens = np.bincount(idx, minlength=40)
    assert int(got[3][0]) == int(sum(lens[q[0]].sum() for q in queries))


def test_buffer_overflow_and_cut(emu, oracle):
    """many more hits than 2*KP = 128 slots: the overflow -> sort -> cut -> threshold path runs several times"""
    rng = np.random.default_rng(7)
    ptr, idx, val = random_csr(rng, 900, 30, 3, 12, neg=True)
    qp, qi, qv = random_csr(rng, 6, 30, 8, 20, neg=True)
    for k, block in ((10, 8), (64, 16), (65, 8), (3, 64)):
        got = run_emu(emu, ptr, idx, val, k, 6, q=(qp, qi, qv), grid=2, block=block)
        queries = rows_of(qp, qi, qv, range(6))
        check(oracle, ptr, idx, val, k, got, queries, [-1] * 6, None)


def test_ties_zero_scores_mask_and_exclude(emu, oracle):
    """equal scores rank by ascending row; zero scores are dropped AFTER the cut to k (they use up slots); mask / exclude"""
    ptr, idx, val, (qp, qi, qv), mask, excl = tie_case()
    for k, expect in TIE_EXPECT.items():
        got = run_emu(emu, ptr, idx, val, k, 3, q=(qp, qi, qv), exclude=excl, mask=mask, grid=1, block=4)
        check(oracle, ptr, idx, val, k, got, rows_of(qp, qi, qv, range(3)), list(excl), mask)
        assert list(got[0][0, :got[2][0]]) == expect
        assert got[2][1] == 0 and got[2][2] == 0


def test_stamps_survive_many_queries_and_launches(emu, oracle):
    """a workgroup's scratch is reused by successive queries and launches without being cleared"""
    rng = np.random.default_rng(11)
    ptr, idx, val = random_csr(rng, 120, 25, 1, 6, zipf=True)
    k = 9
    got = run_emu(emu, ptr, idx, val, k, 120, exclude_self=0, grid=2, block=8, rounds=3, serial_base=4000)
    queries = rows_of(ptr, idx, val, range(120))
    check(oracle, ptr, idx, val, k, got, queries, [-1] * 120, None)


def test_query_subrange_and_large_k(emu, oracle):
    rng = np.random.default_rng(13)
    ptr, idx, val = random_csr(rng, 300, 50, 2, 10)
    k = 200  # KP = 256
    got = run_emu(emu, ptr, idx, val, k, 5, q_first=290, exclude_self=1, grid=5, block=16)
    queries = rows_of(ptr, idx, val, range(290, 295))
    check(oracle, ptr, idx, val, k, got, queries, list(range(290, 295)), None)


# ---- the GPU parity cases themselves, driven through the emulated kernel ------------------------------------------
# tests/test_gpu_vectors_sparse.py cannot run in a container without a GPU; running its cases against the emulation
# checks the test code (expected values, the oracle-free properties) as much as the kernel's control flow.
class EmuSparse:
    """stand-in for gorse_amd.capi.Sparse on top of emu_sparse_search (8 work-items per workgroup)"""
    L = None

    def __init__(self, indptr, indices, values, device=0):
        self.ptr = np.ascontiguousarray(indptr, np.int64)
        self.idx = np.ascontiguousarray(indices, np.uint32)
        self.val = np.ascontiguousarray(values, np.float32)
        self.N = self.ptr.size - 1
        self.mask = None
        self.stats = (0, 0)
        self.launches = 0

    def set_mask(self, admissible=None):
        self.mask = None if admissible is None else np.ascontiguousarray(admissible, np.uint8)

    def _run(self, k, nq, **kw):
        got = run_emu(self.L, self.ptr, self.idx, self.val, k, nq, mask=self.mask, grid=3, block=8,
                      serial_base=1000 * self.launches, **kw)
        self.launches += 1
        self.stats = (int(got[3][0]), int(got[3][1]))
        return got[0], got[1], got[2]

    def search(self, q_indptr, q_indices, q_values, k, exclude=None):
        qp = np.ascontiguousarray(q_indptr, np.int64)
        ex = None if exclude is None else np.ascontiguousarray(exclude, np.int64)
        return self._run(k, qp.size - 1, q=(qp, np.ascontiguousarray(q_indices, np.uint32),
                                            np.ascontiguousarray(q_values, np.float32)), exclude=ex)

    def all_pairs(self, k, q_begin=0, q_end=None, exclude_self=True, fetch=True):
        q_end = self.N if q_end is None else q_end
        return self._run(k, q_end - q_begin, q_first=q_begin, exclude_self=int(bool(exclude_self)))

    def last_stats(self):
        return self.stats

    def set_profiling(self, on):
        pass

    def get_profile(self):
        return 1, 1.0


def _gpu_cases():
    import test_gpu_vectors_sparse as G
    cases = []
    for name in sorted(dir(G)):
        if not name.startswith("test_") or name in ("test_argument_errors", "test_reference_suite_sparse",
                                                    "test_stamp_counter_wraps_by_clearing_the_scratch",
                                                    "test_postings_built_on_the_device_give_the_same_results",
                                                    "test_heavy_queries_take_the_row_streaming_path",
                                                    "test_random_configurations", "test_edge_inputs"):
            continue  # these drive sparse.hip itself (tests/test_sparse_fake_runtime_cpu.py runs them); the database suite
            # has its own emulated run below
        if name == "test_many_hits_overflow_the_ranking_buffer":
            cases += [(name, k) for k in (3, 64, 65, 513)]
        else:
            cases.append((name, None))
    return cases


@pytest.mark.parametrize("name,k", _gpu_cases())
def test_gpu_case_on_the_emulated_kernel(emu, oracle, monkeypatch, name, k):
    import test_gpu_vectors_sparse as G
    EmuSparse.L = emu
    monkeypatch.setattr(G.capi, "Sparse", EmuSparse)
    fn = getattr(G, name)
    args = {"oracle": oracle, "k": k}
    fn(**{a: args[a] for a in fn.__code__.co_varnames[:fn.__code__.co_argcount]})


def test_no_out_of_bounds_access_under_address_sanitizer(tmp_path):
    """the same driver under AddressSanitizer + UndefinedBehaviorSanitizer: an index past the LDS ranking buffer, the
    scratch rows, the posting arrays or the result rows is a report"""
    exe = str(tmp_path / "asan_sparse")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                            "-ffp-contract=off", "-pthread", os.path.join(EMU, "tsan_main.cpp"), os.path.join(EMU, "sparse_emu.cpp"),
                            "-o", exe], capture_output=True, text=True)
    if build.returncode != 0:
        pytest.skip("AddressSanitizer is not available to this g++: " + build.stderr[-200:])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert "Sanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-2000:]
    assert run.returncode == 0 and "tsan run done rc=0" in run.stdout


def test_no_data_race_between_barriers_under_thread_sanitizer(tmp_path):
    """tests/emu/tsan_main.cpp: the emulated kernel under ThreadSanitizer -- two work-items reaching the same plain
    load/store without a barrier in between is a reported race (removing the barrier after a posting list gives four
    reports), so a clean run vouches for the barrier placement of sparse_kernels.hpp"""
    exe = str(tmp_path / "tsan_sparse")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-ffp-contract=off", "-pthread",
                            os.path.join(EMU, "tsan_main.cpp"), os.path.join(EMU, "sparse_emu.cpp"), "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0:
        pytest.skip("ThreadSanitizer is not available to this g++: " + build.stderr[-200:])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert "ThreadSanitizer" not in run.stderr, run.stderr[-2000:]
    assert run.returncode == 0 and "tsan run done rc=0" in run.stdout


# ---- the vectors.Database twin and the sparse logics kinds with the emulated kernel as their searcher -----------------
def _suite_cases():
    import vectors_suite as S
    return S.SPARSE_CASES


@pytest.mark.parametrize("case", _suite_cases(), ids=lambda f: f.__name__)
def test_database_suite_on_the_emulated_kernel(emu, case):
    from gorse_amd import vectors as V

    def sparse_searcher(n, indptr, indices, values, admissible, nq, q_indptr, q_indices, q_values, k, idx, score, cnt):
        p = lambda x: C.cast(x, C.c_void_p)
        return emu.emu_sparse_search(n, p(indptr), p(indices), p(values), nq, p(q_indptr), p(q_indices), p(q_values), 0, None, 0,
                                     p(admissible), k, 2, 8, 1, 0, p(idx), p(score), p(cnt), None)
    case(V.Database(sparse_searcher=sparse_searcher))


def test_random_database_operations_on_the_emulated_kernel(emu, oracle):
    """tests/test_vectors_db_cpu.py's random-operations test with the emulated kernel behind the sparse collection (the dense
    one keeps the oracle-backed searcher)"""
    import test_vectors_db_cpu as T
    from gorse_amd import vectors as V

    def dense_searcher(X, n, d, metric, Q, nq, k, idx, dist, cnt):
        Xa, Qa = np.ctypeslib.as_array(X, (n, d)).copy(), np.ctypeslib.as_array(Q, (nq, d)).copy()
        I, D, Cn = np.ctypeslib.as_array(idx, (nq, k)), np.ctypeslib.as_array(dist, (nq, k)), np.ctypeslib.as_array(cnt, (nq,))
        for t in range(nq):
            ei, ed = oracle.search_vector(Xa, metric, Qa[t], k)
            I[t, :ei.size], D[t, :ei.size], Cn[t] = ei, ed, ei.size
        return 0

    def sparse_searcher(n, indptr, indices, values, admissible, nq, q_indptr, q_indices, q_values, k, idx, score, cnt):
        p = lambda x: C.cast(x, C.c_void_p)
        return emu.emu_sparse_search(n, p(indptr), p(indices), p(values), nq, p(q_indptr), p(q_indices), p(q_values), 0, None, 0,
                                     p(admissible), k, 2, 8, 1, 0, p(idx), p(score), p(cnt), None)
    T.test_random_operations_against_a_model_of_the_reference(V.Database(searcher=dense_searcher, sparse_searcher=sparse_searcher))


@pytest.mark.parametrize("block", [1, 7, 64])
def test_postings_built_by_the_device_kernels(emu, oracle, block):
    """sparse_count_kernel / sparse_scan_kernel / sparse_scatter_kernel under emulation: the same posting directory as the
    host build, every list the same set of (row, value) entries, and queries over the device-built lists (whose entries
    come in the atomics' order) still equal the oracle bit for bit.  Empty rows, empty lists and a single row included."""
    rng = np.random.default_rng(40 + block)
    emu.emu_sparse_set_device_build(1)
    try:
        for rows, dims, lo, hi in ((90, 37, 0, 8), (1, 5, 3, 3), (40, 300, 0, 3)):
            ptr, idx, val = random_csr(rng, rows, dims, lo, hi, neg=True, zipf=rows > 1)
            got = run_emu(emu, ptr, idx, val, 6, rows, exclude_self=1, grid=2, block=block)  # asserts rc == 0
            check(oracle, ptr, idx, val, 6, got, rows_of(ptr, idx, val, range(rows)), list(range(rows)))
    finally:
        emu.emu_sparse_set_device_build(0)


@pytest.mark.parametrize("block", [1, 8, 64])
def test_heavy_queries_by_row_streaming(emu, oracle, block):
    """sparse_heavy_score_kernel + sparse_heavy_rank_kernel under emulation: queries with more than 5 entries are merged against
    every stored row instead of walking posting lists; light and heavy queries in one call, three batches of heavy ones"""
    rng = np.random.default_rng(60 + block)
    ptr, idx, val = random_csr(rng, 150, 40, 0, 12, neg=True, zipf=True)
    assert 16 < int((np.diff(ptr) > 5).sum()) < 140
    mask = (rng.random(150) < 0.8).astype(np.uint8)
    emu.emu_sparse_set_heavy(5)
    try:
        for k in (4, 70):
            got = run_emu(emu, ptr, idx, val, k, 150, exclude_self=1, mask=mask, grid=3, block=block)
            check(oracle, ptr, idx, val, k, got, rows_of(ptr, idx, val, range(150)), list(range(150)), mask)
    finally:
        emu.emu_sparse_set_heavy(0)


@pytest.mark.parametrize("block", [4, 64])
def test_hot_rows_in_lds(emu, oracle, block):
    """sparse_query_kernel<KP, 512>: the accumulators of the 512 longest stored rows live in LDS, the others in the global
    scratch row; both kinds in every query, stamps carried across the queries of a workgroup and across launches"""
    rng = np.random.default_rng(70 + block)
    ptr, idx, val = random_csr(rng, 1300, 90, 0, 9, neg=True, zipf=True)
    mask = (rng.random(1300) < 0.85).astype(np.uint8)
    emu.emu_sparse_set_hot(512)
    try:
        got = run_emu(emu, ptr, idx, val, 9, 140, q_first=600, exclude_self=1, mask=mask, grid=3, block=block, rounds=2)
        check(oracle, ptr, idx, val, 9, got, rows_of(ptr, idx, val, range(600, 740)), list(range(600, 740)), mask)
        small = random_csr(rng, 40, 12, 1, 5)  # fewer rows than LDS cells
        got = run_emu(emu, *small, 5, 40, exclude_self=0, grid=2, block=block)
        check(oracle, *small, 5, got, rows_of(*small, range(40)), [-1] * 40)
    finally:
        emu.emu_sparse_set_hot(0)
