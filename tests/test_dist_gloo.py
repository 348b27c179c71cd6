"""The N > 1 path on CPU: world_size 2 over gloo.  The sharding / exchange logic (gorse_amd.dist) is the
code bench.py runs on GPUs; here the per-rank compute engine is the CPU oracle (tests may use it), and the
result must equal a single-process emulation of the same schedule, exactly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gorse_amd import dist as gdist  # noqa: E402
from gorse_amd import synth  # noqa: E402


class OracleEngine:
    """CPU stand-in for HipEngine: the oracle's sequential BPR epoch on this rank's user shard."""

    def __init__(self, P, Q, uptr, uidx):
        from oracle import oracle as orc
        self.o = orc.Oracle()
        self.P, self.Q = P.copy(), Q.copy()
        self.Qsync = Q.copy()
        self.uptr, self.uidx = uptr, uidx
        self.srt = orc.sort_rows(uptr, uidx)

    def epoch(self, n, lr, reg, seed, epoch, base):
        self.o.bpr_epoch_sampled(self.P, self.Q, self.uptr, self.uidx, self.srt, seed, epoch, base, n, lr, reg)

    def export_delta(self):
        return torch.from_numpy((self.Q - self.Qsync).ravel().copy())

    def import_delta(self, delta):
        self.Q = (self.Qsync + delta.numpy().reshape(self.Q.shape)).astype(np.float32)
        self.Qsync = self.Q.copy()


def _problem():
    data = synth.synth_cf(120, 80, 2400, seed=3, min_len=3, with_test=False)
    P, Q = synth.init_factors(data.U, data.I, 16, 0.0, 0.1, 1)
    return data, P, Q


def _rank_inputs(data, P, rank, world):
    lo, hi = gdist.shard_range(data.U, rank, world)
    uptr, uidx = gdist.shard_csr(data.uptr, data.uidx, lo, hi)
    active = int((np.diff(uptr) > 0).sum())
    total_active = int((np.diff(data.uptr) > 0).sum())
    n = gdist.samples_for_rank(data.n_train, active, total_active)
    return lo, hi, uptr, uidx, n


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data, P, Q = _problem()
    lo, hi, uptr, uidx, n = _rank_inputs(data, P, rank, world)
    eng = OracleEngine(P[lo:hi], Q, uptr, uidx)
    comm = gdist.TorchComm()
    for ep in range(1, 4):
        gdist.run_epoch(eng, comm, n, 0.05, 0.01, 11, ep, rank * (1 << 40))
    np.save(os.path.join(out, "P%d.npy" % rank), eng.P)
    np.save(os.path.join(out, "Q%d.npy" % rank), eng.Q)
    dist.destroy_process_group()


def test_shard_range_partitions_rows():
    for n in (1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            spans = [gdist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
            tiled = [gdist.shard_range(n, r, w, align=128) for r in range(w)]
            assert tiled[0][0] == 0 and tiled[-1][1] == n and all(a[1] == b[0] for a, b in zip(tiled, tiled[1:]))
            if n // w >= 8 * 128:  # shards much larger than a tile: every boundary on a tile, within half a tile of the even split
                assert all(l % 128 == 0 for l, _ in tiled) and all(abs(a[0] - b[0]) <= 64 for a, b in zip(tiled, spans))
            else:  # small shards keep the even split (no empty or doubled shard: 200 rows over 4 ranks)
                assert tiled == spans
    assert [gdist.shard_range(200, r, 4, align=128) for r in range(4)] == [(0, 50), (50, 100), (100, 150), (150, 200)]
    assert [gdist.shard_range(1_000_000, r, 8, align=128) for r in range(8)][3] == (375040, 499968)


def test_two_rank_bpr_matches_single_process_emulation(tmp_path):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    # emulation: both shards run from the same Q, deltas are summed, per epoch
    data, P, Q = _problem()
    engines = []
    for r in range(world):
        lo, hi, uptr, uidx, n = _rank_inputs(data, P, r, world)
        engines.append((OracleEngine(P[lo:hi], Q, uptr, uidx), n))
    for ep in range(1, 4):
        deltas = []
        for r, (e, n) in enumerate(engines):
            e.epoch(n, 0.05, 0.01, 11, ep, r * (1 << 40))
            deltas.append(e.export_delta())
        total = deltas[0] + deltas[1]
        for e, _ in engines:
            e.import_delta(total.clone())
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("P%d.npy" % r)), engines[r][0].P)
        assert np.array_equal(np.load(tmp_path / ("Q%d.npy" % r)), engines[r][0].Q)
    # every rank ends with the same replica of Q
    assert np.array_equal(np.load(tmp_path / "Q0.npy"), np.load(tmp_path / "Q1.npy"))
    assert not np.array_equal(np.load(tmp_path / "Q0.npy"), Q)


# ---- row-sharded ALS (gorse_amd.dist.run_als_epoch) -----------------------------------------------------------
class OracleAlsEngine:
    """CPU stand-in for HipAlsEngine: the oracle's half-sweep on this rank's row ranges, full replicas of P and Q."""

    def __init__(self, data, P, Q, rank, world):
        from oracle import oracle as orc
        self.o = orc.Oracle()
        self.data, self.rank, self.world = data, rank, world
        self.F = [P.copy(), Q.copy()]
        self.rows = (data.U, data.I)
        self.range = [gdist.shard_range(n, rank, world) for n in self.rows]
        self.block = [gdist.block_rows(n, world) for n in self.rows]

    def half(self, side, weight, reg):
        d = self.data
        ptr, idx, bptr = (d.uptr, d.uidx, d.iptr) if side == 0 else (d.iptr, d.iidx, d.uptr)
        lo, hi = self.range[side]
        self.o.als_half_range(self.F[side], self.F[1 - side], ptr, idx, bptr, weight, reg, lo, hi)

    def export_block(self, side):
        lo, hi = self.range[side]
        buf = np.zeros((self.block[side], self.F[side].shape[1]), np.float32)
        buf[:hi - lo] = self.F[side][lo:hi]
        return torch.from_numpy(buf.ravel())

    def import_blocks(self, side, gathered):
        g = gathered.numpy().reshape(self.world, self.block[side], -1)
        for r in range(self.world):
            lo, hi = gdist.shard_range(self.rows[side], r, self.world)
            if r != self.rank:
                self.F[side][lo:hi] = g[r, :hi - lo]


def _als_problem():
    data = synth.synth_cf(101, 67, 1500, seed=5, min_len=2, with_test=False)  # odd sizes: uneven shards
    P, Q = synth.init_factors(data.U, data.I, 16, 0.0, 0.1, 2)
    return data, P, Q


def _als_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data, P, Q = _als_problem()
    eng = OracleAlsEngine(data, P, Q, rank, world)
    comm = gdist.TorchComm()
    for _ in range(3):
        gdist.run_als_epoch(eng, comm, 0.05, 0.015)
    np.save(os.path.join(out, "alsP%d.npy" % rank), eng.F[0])
    np.save(os.path.join(out, "alsQ%d.npy" % rank), eng.F[1])
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_als_equals_the_single_process_epoch(tmp_path, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_als_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as orc
    o = orc.Oracle()
    data, P, Q = _als_problem()
    eP, eQ = P, Q
    for _ in range(3):
        eP, eQ = o.als_epoch(eP, eQ, data.uptr, data.uidx, data.iptr, data.iidx, 0.05, 0.015)
    for r in range(world):  # rows are independent inside a half-sweep: sharding changes nothing, bit for bit
        assert np.array_equal(np.load(tmp_path / ("alsP%d.npy" % r)), eP)
        assert np.array_equal(np.load(tmp_path / ("alsQ%d.npy" % r)), eQ)


# ---- sharded Evaluate (gorse_amd.dist.evaluate_sharded) ---------------------------------------------------------
class OracleEvalEngine:
    """CPU stand-in for HipEngine.eval_partial: rank lists from the oracle's Rank, metric arithmetic and partial sums from
    gorse_amd.metrics (the code the GPU path runs on the host)."""

    def __init__(self, P_local, Q, test_ptr, test_idx, neg_ptr, neg_idx):
        from oracle import oracle as orc
        self.o = orc.Oracle()
        self.P, self.Q = P_local, Q
        self.split = (test_ptr, test_idx, neg_ptr, neg_idx)

    def eval_partial(self, topk, metrics):
        from gorse_amd import metrics as M
        test_ptr, test_idx, neg_ptr, neg_idx = self.split
        users = np.nonzero(np.diff(test_ptr) > 0)[0].astype(np.int32)
        cptr, cidx = M.candidates(test_ptr, test_idx, neg_ptr, neg_idx, users)
        rank, rlen = self.o.mf_rank(self.P, self.Q, users, cptr, cidx, topk)
        return M.partial_sums(rank, rlen, users, test_ptr, test_idx, metrics)


def _eval_problem():
    data = synth.synth_cf(151, 90, 3000, seed=8, min_len=3, n_neg=30)
    P, Q = synth.init_factors(data.U, data.I, 16, 0.0, 0.3, 6)
    return data, P, Q


def _eval_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data, P, Q = _eval_problem()
    lo, hi = gdist.shard_range(data.U, rank, world)
    tp, ti = gdist.shard_csr(data.test_ptr, data.test_idx, lo, hi)
    npr, ni = gdist.shard_csr(data.neg_ptr, data.neg_idx, lo, hi)
    eng = OracleEvalEngine(P[lo:hi], Q, tp, ti, npr, ni)
    score = gdist.evaluate_sharded(eng, gdist.TorchComm(), 10)
    np.save(os.path.join(out, "eval%d.npy" % rank), score)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_evaluate_equals_evaluate(tmp_path, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_eval_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as orc
    data, P, Q = _eval_problem()
    ref = orc.Oracle().evaluate(P, Q, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)
    for r in range(world):  # every rank holds the same NDCG / Precision / Recall; worker-partial sums as in the reference
        got = np.load(tmp_path / ("eval%d.npy" % r))
        assert np.allclose(got, ref, atol=2e-6), (got, ref)
    assert np.array_equal(np.load(tmp_path / "eval0.npy"), np.load(tmp_path / ("eval%d.npy" % (world - 1))))


# ---- the GPU engines themselves (HipEngine / HipAlsEngine) over gloo, with a stand-in for the gorse_mf handle ----------
class StandInMF:
    """Implements the gorse_mf calls the engines make (capi.MF's method names and pointer conventions) with the oracle
    as the compute and host memory as "device" memory, so that the engines' own code -- ranges, block sizes, buffer
    offsets, call order -- runs under gloo exactly as it runs under RCCL."""

    def __init__(self, data, P, Q):
        import ctypes
        from oracle import oracle as orc
        self.ct, self.o = ctypes, orc.Oracle()
        self.data = data
        self.U, self.I, self.d = P.shape[0], Q.shape[0], P.shape[1]
        self.F = [np.ascontiguousarray(P.copy()), np.ascontiguousarray(Q.copy())]
        self.Qsync = None
        self.ranges = [(0, self.U), (0, self.I)]
        self.srt = orc.sort_rows(data.uptr, data.uidx) if data is not None else None

    def _view(self, ptr, n):
        return np.ctypeslib.as_array(self.ct.cast(ptr, self.ct.POINTER(self.ct.c_float)), (n,))

    # ALS
    def als_set_ranges(self, u0, u1, i0, i1):
        self.ranges = [(u0, u1), (i0, i1)]

    def als_half_epoch(self, side, w, reg):
        d = self.data
        ptr, idx, bptr = (d.uptr, d.uidx, d.iptr) if side == 0 else (d.iptr, d.iidx, d.uptr)
        self.o.als_half_range(self.F[side], self.F[1 - side], ptr, idx, bptr, w, reg, *self.ranges[side])

    als_half_epoch_enqueue = als_half_epoch  # the stand-in has no stream: "enqueue" runs it

    def rows_export(self, side, lo, hi, ptr):
        self._view(ptr, (hi - lo) * self.d)[:] = self.F[side][lo:hi].ravel()

    def rows_import(self, side, lo, hi, ptr):
        self.F[side][lo:hi] = self._view(ptr, (hi - lo) * self.d).reshape(hi - lo, self.d)

    # BPR
    def bpr_epoch_enqueue(self, n, lr, reg, seed, epoch, sample_base=0, mode=0):
        self.o.bpr_epoch_sampled(self.F[0], self.F[1], self.data.uptr, self.data.uidx, self.srt, seed, epoch, sample_base, n, lr, reg)

    def item_sync_mark(self):
        self.Qsync = self.F[1].copy()

    def synchronize(self):
        pass

    def item_delta_export(self, ptr):
        self._view(ptr, self.I * self.d)[:] = (self.F[1] - self.Qsync).ravel()

    def item_delta_import(self, ptr):
        self.F[1] = (self.Qsync + self._view(ptr, self.I * self.d).reshape(self.I, self.d)).astype(np.float32)
        self.Qsync = self.F[1].copy()


def _hip_engines_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = gdist.TorchComm()
    # ALS through HipAlsEngine
    data, P, Q = _als_problem()
    eng = gdist.HipAlsEngine(StandInMF(data, P, Q), rank, world, device="cpu")
    for _ in range(2):
        gdist.run_als_epoch(eng, comm, 0.05, 0.015)
    np.save(os.path.join(out, "hP%d.npy" % rank), eng.mf.F[0])
    np.save(os.path.join(out, "hQ%d.npy" % rank), eng.mf.F[1])
    # BPR through HipEngine
    data, P, Q = _problem()
    lo, hi, uptr, uidx, n = _rank_inputs(data, P, rank, world)
    import types
    shard = types.SimpleNamespace(uptr=uptr, uidx=uidx)
    beng = gdist.HipEngine(StandInMF(shard, P[lo:hi], Q), 0, device="cpu")
    beng.enable_exchange()
    for ep in range(1, 3):
        gdist.run_epoch(beng, comm, n, 0.05, 0.01, 11, ep, rank * (1 << 40))
    np.save(os.path.join(out, "bP%d.npy" % rank), beng.mf.F[0])
    np.save(os.path.join(out, "bQ%d.npy" % rank), beng.mf.F[1])
    dist.destroy_process_group()


def test_hip_engines_over_gloo_with_a_stand_in_handle(tmp_path):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_hip_engines_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle as orc
    o = orc.Oracle()
    data, P, Q = _als_problem()
    eP, eQ = P, Q
    for _ in range(2):
        eP, eQ = o.als_epoch(eP, eQ, data.uptr, data.uidx, data.iptr, data.iidx, 0.05, 0.015)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("hP%d.npy" % r)), eP) and np.array_equal(np.load(tmp_path / ("hQ%d.npy" % r)), eQ)
    # BPR: the same emulation as test_two_rank_bpr_matches_single_process_emulation, two epochs
    data, P, Q = _problem()
    engines = []
    for r in range(world):
        lo, hi, uptr, uidx, n = _rank_inputs(data, P, r, world)
        engines.append((OracleEngine(P[lo:hi], Q, uptr, uidx), n))
    for ep in range(1, 3):
        deltas = []
        for r, (e, n) in enumerate(engines):
            e.epoch(n, 0.05, 0.01, 11, ep, r * (1 << 40))
            deltas.append(e.export_delta())
        total = deltas[0] + deltas[1]
        for e, _ in engines:
            e.import_delta(total.clone())
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("bP%d.npy" % r)), engines[r][0].P)
        assert np.array_equal(np.load(tmp_path / ("bQ%d.npy" % r)), engines[r][0].Q)


# ---- one process, N handles: the Go master's mode (integration/go/model/cf/rccl_hip.go) -----------------------------------------
class StubLocalComms:
    """gorse_mf_item_allreduce / gorse_mf_rows_allgather over N stand-in handles of ONE process: what the grouped RCCL calls of
    csrc/comm.hip leave in the replicas (export kernel -> all-reduce -> import kernel; one broadcast per owner)."""

    def __init__(self, world):
        self.world = world
        self.calls = []

    def item_allreduce(self, mfs):
        assert len(mfs) == self.world
        self.calls.append("item_allreduce x%d" % len(mfs))
        total = sum((m.F[1] - m.Qsync) for m in mfs)  # rank order, like the emulation the multi-process test compares with
        for m in mfs:
            m.F[1] = (m.Qsync + total).astype(np.float32)
            m.Qsync = m.F[1].copy()

    def rows_allgather(self, mfs, side, row_splits):
        assert len(mfs) == self.world and row_splits[0] == 0 and len(row_splits) == self.world + 1
        self.calls.append("rows_allgather side %d" % side)
        for r, owner in enumerate(mfs):
            lo, hi = row_splits[r], row_splits[r + 1]
            for m in mfs:
                if m is not owner:
                    m.F[side][lo:hi] = owner.F[side][lo:hi]


@pytest.mark.parametrize("world", [2, 3])
def test_one_process_n_handles_call_sequence(world):
    """run_epoch_local / run_als_epoch_local (the Python twins of hipGroup.bprEpoch / alsEpochSharded): N handles driven from one
    thread -- every device's epoch enqueued, then ONE grouped exchange -- give exactly what one process per rank gives: the
    single-process emulation for BPR, the unsharded oracle epoch for ALS."""
    import types
    from oracle import oracle as orc
    data, P, Q = _problem()
    engines, samples, ref = [], [], []
    for r in range(world):
        lo, hi, uptr, uidx, n = _rank_inputs(data, P, r, world)
        mf = StandInMF(types.SimpleNamespace(uptr=uptr, uidx=uidx), P[lo:hi], Q)
        mf.item_sync_mark()
        engines.append(gdist.HipEngine(mf, 0, device="cpu"))
        samples.append(n)
        ref.append((OracleEngine(P[lo:hi], Q, uptr, uidx), n))
    comms = StubLocalComms(world)
    for ep in range(1, 4):
        gdist.run_epoch_local(engines, comms, samples, 0.05, 0.01, 11, ep)
        deltas = []
        for r, (e, n) in enumerate(ref):
            e.epoch(n, 0.05, 0.01, 11, ep, r * (1 << 40))
            deltas.append(e.export_delta())
        total = sum(deltas[1:], deltas[0])
        for e, _ in ref:
            e.import_delta(total.clone())
    assert comms.calls == ["item_allreduce x%d" % world] * 3  # one grouped call per epoch
    for r in range(world):
        assert np.array_equal(engines[r].mf.F[0], ref[r][0].P) and np.array_equal(engines[r].mf.F[1], ref[r][0].Q)
    # ALS: every handle holds the whole data set and solves its row ranges
    data, P, Q = _als_problem()
    aengines = [gdist.HipAlsEngine(StandInMF(data, P, Q), r, world, device="cpu", staging=False) for r in range(world)]
    comms = StubLocalComms(world)
    eP, eQ = P, Q
    o = orc.Oracle()
    for _ in range(2):
        gdist.run_als_epoch_local(aengines, comms, 0.05, 0.015)
        eP, eQ = o.als_epoch(eP, eQ, data.uptr, data.uidx, data.iptr, data.iidx, 0.05, 0.015)
    assert comms.calls == ["rows_allgather side 0", "rows_allgather side 1"] * 2
    for e in aengines:
        assert np.array_equal(e.mf.F[0], eP) and np.array_equal(e.mf.F[1], eQ)


# ---- the neighbour refresh (sparse item-to-item / user-to-user, dense top-k) over row shards --------------------------------
class StandInSparse:
    """capi.Sparse's all_pairs with the oracle as the compute (tests only)"""

    def __init__(self, ptr, idx, val):
        from oracle import oracle as orc
        self.o, self.ptr, self.idx, self.val, self.N = orc.Oracle(), ptr, idx, val, ptr.size - 1

    def all_pairs(self, k, q_begin=0, q_end=None):
        q_end = self.N if q_end is None else q_end
        n = q_end - q_begin
        I, S, Cn = np.full((n, k), -1, np.int32), np.full((n, k), -np.inf, np.float32), np.zeros(n, np.int32)
        for t, q in enumerate(range(q_begin, q_end)):
            ei, es = self.o.sparse_search(self.ptr, self.idx, self.val, self.idx[self.ptr[q]:self.ptr[q + 1]],
                                          self.val[self.ptr[q]:self.ptr[q + 1]], k, exclude=q)
            I[t, :ei.size], S[t, :ei.size], Cn[t] = ei, es, ei.size
        return I, S, Cn


class StandInTopK:
    """capi.TopK's all_pairs (SearchIndex for every stored vector: distances ascending, padded with -1 / +inf) with the
    oracle as the compute (tests only)"""

    def __init__(self, X, metric):
        from oracle import oracle as orc
        self.o, self.X, self.metric, self.N = orc.Oracle(), X, metric, X.shape[0]

    def all_pairs(self, k, q_begin=0, q_end=None):
        q_end = self.N if q_end is None else q_end
        I = np.full((q_end - q_begin, k), -1, np.int32)
        D = np.full((q_end - q_begin, k), np.inf, np.float32)
        for t, q in enumerate(range(q_begin, q_end)):
            ei, ed = self.o.search_index(self.X, self.metric, q, k)
            I[t, :ei.size], D[t, :ei.size] = ei, ed
        return I, D


def _i2i_problem():
    data = synth.synth_cf(400, 53, 500, seed=21, min_len=1, with_test=False)  # 53 items: shards of unequal size
    return synth.idf_vectors(data.iptr, data.iidx, data.U)


def _refresh_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ptr, idx, val = _i2i_problem()
    eng = gdist.HipNeighborsEngine(StandInSparse(ptr, idx, val), device="cpu")
    I, S, Cn = gdist.refresh_neighbors_sharded(eng, gdist.TorchComm(), 12)
    # the same refresh over a dense index (embedding item-to-item): 23 vectors, k = 30 > N - 1, so every row is padded
    X = np.random.default_rng(5).standard_normal((23, 8)).astype(np.float32)
    dI, dD, dC = gdist.refresh_neighbors_sharded(gdist.HipNeighborsEngine(StandInTopK(X, 1), device="cpu"), gdist.TorchComm(), 30)
    np.savez(os.path.join(out, "nb%d.npz" % rank), I=I, S=S, Cn=Cn, dI=dI, dD=dD, dC=dC)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_neighbor_refresh_equals_the_single_process_result(tmp_path, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_refresh_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ptr, idx, val = _i2i_problem()
    I, S, Cn = StandInSparse(ptr, idx, val).all_pairs(12)
    assert Cn.max() == 12 and Cn.min() < 12  # full and padded rows both occur
    for r in range(world):
        got = np.load(tmp_path / ("nb%d.npz" % r))
        assert np.array_equal(got["I"], I) and np.array_equal(got["S"].view(np.uint32), S.view(np.uint32))
        assert np.array_equal(got["Cn"], Cn)
    X = np.random.default_rng(5).standard_normal((23, 8)).astype(np.float32)
    dI, dD = StandInTopK(X, 1).all_pairs(30)
    for r in range(world):
        got = np.load(tmp_path / ("nb%d.npz" % r))
        assert np.array_equal(got["dI"], dI) and np.array_equal(got["dD"].view(np.uint32), dD.view(np.uint32))
        assert (got["dC"] == 22).all()


def _libcomm_worker(rank, world, port, out):
    """every rank must come out of LibComm's constructor -- with a communicator or with an exception -- whatever happens on
    rank 0: here there is no GPU, so the library either cannot draw an id (rank 0) or cannot create the communicator"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def share(uid):
        t = torch.tensor(list(uid), dtype=torch.uint8)
        dist.broadcast(t, src=0)
        return bytes(t.tolist())
    outcome = "communicator"
    try:
        gdist.LibComm(rank, world, 0, share)
    except Exception as e:
        outcome = "raised %s" % type(e).__name__
    # the agreement step of bench.py's make_comm: reached by every rank
    ok = torch.tensor([1.0 if outcome == "communicator" else 0.0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    with open(os.path.join(out, "rank%d.txt" % rank), "w") as f:
        f.write("%s|%g" % (outcome, float(ok.item())))
    dist.destroy_process_group()


def test_library_communicator_setup_never_leaves_a_rank_behind(tmp_path):
    """bench.py's multi-rank path sets the library's RCCL communicator up behind a torch broadcast of rank 0's id; a rank that
    failed before the broadcast used to leave the others waiting in it.  Without a GPU every rank fails -- and every rank must
    reach the agreement step and take the torch.distributed fallback together."""
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_libcomm_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read() for r in range(world)]
    assert all(g.startswith("raised") and g.endswith("|0") for g in got), got


# ---- the triangle-sharded dense all-pairs search: protocol + exchange over gloo with a stand-in engine ---------------------------
class NumpyTriEngine:
    """Stands in for a gorse_topk handle behind gorse_amd.dist.refresh_neighbors_triangle (what HipTriEngine drives on a GPU: include/
    gorse_hip.h gorse_topk_tri_*): exact -dot scores in numpy, the SAME ownership, slices and messages -- rank r runs the pilots of its
    slice (threshold = k-th best over every 4th row: a lower bound of the true k-th best), sweeps the query blocks r, r + world, ...
    (own lists: rows of the blocks up to its own; foreign lists: for the rows of every EARLIER block, its columns that reach the row's
    threshold), packs per destination a count per owned query + (key, row) entries, and ranks own + foreign candidates at the end."""
    BLOCK = 512

    def __init__(self, X):
        self.X, self.n_rows = X.astype(np.float64), X.shape[0]
        self.S = -(self.X @ self.X.T)  # distances: -dot
        np.fill_diagonal(self.S, np.inf)

    def _blocks(self, r):
        return list(range(r, -(-self.n_rows // self.BLOCK), self.world))

    def _rows_of(self, c):
        return np.arange(c * self.BLOCK, min((c + 1) * self.BLOCK, self.n_rows))

    def begin(self, k, rank, world):
        self.k, self.rank, self.world = k, rank, world
        self.thr = np.full(self.n_rows, np.nan)
        lo, hi, _ = self.slice(rank)
        sample = np.arange(0, self.n_rows, 4)
        for q in range(lo, hi):
            self.thr[q] = np.sort(self.S[q, sample])[k]  # (k + 1)-th smallest of a sample: >= the true k-th smallest distance
        self.own, self.foreign = {}, {q: [] for q in range(self.n_rows)}

    def slice(self, r):
        nblk = -(-self.n_rows // self.BLOCK)
        lo, hi = nblk * r // self.world * self.BLOCK, nblk * (r + 1) // self.world * self.BLOCK
        owned = sum(self._rows_of(c).size for c in self._blocks(r))
        return min(lo, self.n_rows), min(hi, self.n_rows), owned

    def thresholds(self):
        lo, hi, _ = self.slice(self.rank)
        return self.thr[lo:hi].astype(np.float64)

    def put_thresholds(self, lo, hi, a):
        self.thr[lo:hi] = a

    def sweep(self):
        assert not np.isnan(self.thr).any()  # every rank's slice has arrived
        for c in self._blocks(self.rank):
            cols = self._rows_of(c)
            upto = min((c + 1) * self.BLOCK, self.n_rows)
            for q in cols:  # own lists: the rows of the blocks 0 .. c
                r = np.flatnonzero(self.S[q, :upto] <= self.thr[q])
                self.own[q] = list(zip(self.S[q, r].tolist(), r.tolist()))
            for r in range(c * self.BLOCK):  # the earlier blocks' rows as queries: candidates among this block's columns
                hit = cols[self.S[r, cols] <= self.thr[r]]
                self.foreign[r].extend(zip(self.S[r, hit].tolist(), hit.tolist()))

    def _owned(self, r):
        return np.concatenate([self._rows_of(c) for c in self._blocks(r)]) if self._blocks(r) else np.zeros(0, np.int64)

    def pack(self, dest):
        qs = self._owned(dest)
        counts = np.array([len(self.foreign[q]) for q in qs], np.int32)
        ent = [np.float32(d_).view(np.uint32).astype(np.uint64) << np.uint64(32) | np.uint64(i) for q in qs for d_, i in self.foreign[q]]
        return counts, np.array(ent, np.uint64).view(np.int64)  # the entries travel as 64-bit words

    def unpack(self, src, counts, entries):
        entries = np.asarray(entries).view(np.uint64)
        qs, at = self._owned(self.rank), 0
        assert counts.size == qs.size and int(counts.sum()) == entries.size
        for q, c in zip(qs, counts):
            for e in entries[at:at + c]:
                self.foreign[q].append((float(np.uint32(e >> np.uint64(32)).view(np.float32)), int(e & np.uint64(0xffffffff))))
            at += c

    def finish(self, idx=None, dist=None):
        if idx is None:
            idx, dist = np.full((self.n_rows, self.k), -1, np.int32), np.full((self.n_rows, self.k), np.inf, np.float32)
        for q in self._owned(self.rank):
            cand = sorted(set((np.float32(d_), i) for d_, i in self.own[q] + self.foreign[q]))[:self.k]
            idx[q], dist[q] = [i for _, i in cand], [d_ for d_, _ in cand]
        return idx, dist

    def synchronize(self):
        pass


def _tri_problem():
    rng = np.random.default_rng(17)
    return rng.standard_normal((1300, 12)).astype(np.float32), 7  # three blocks, the last one partial


def _tri_worker(rank, world, port, out):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    X, k = _tri_problem()
    idx, dst = gdist.refresh_neighbors_triangle(NumpyTriEngine(X), gdist.TorchComm(), k)
    np.save(os.path.join(out, "ti%d.npy" % rank), idx)
    np.save(os.path.join(out, "td%d.npy" % rank), dst)
    dist.destroy_process_group()


def test_triangle_sharded_search_protocol_over_gloo(tmp_path):
    """refresh_neighbors_triangle over a world of 2 and 3 (gloo): thresholds all-gathered, foreign lists through the all-to-all of
    variable-size messages, every rank ends with ALL rows, equal to the brute force; and the one-process form
    (refresh_neighbors_triangle_local: what emulates the ranks on one GPU) gives the same rows."""
    X, k = _tri_problem()
    S = -(X.astype(np.float64) @ X.astype(np.float64).T)
    np.fill_diagonal(S, np.inf)
    want = np.argsort(S, axis=1, kind="stable")[:, :k]
    for world in (2, 3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_tri_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
        for r in range(world):
            assert np.array_equal(np.load(tmp_path / ("ti%d.npy" % r)), want), (world, r)
        tm = {}
        idx, _ = gdist.refresh_neighbors_triangle_local([NumpyTriEngine(X) for _ in range(world)], k, timings=tm)
        assert np.array_equal(idx, want) and len(tm["message_bytes"]) == world * (world - 1)
    # ownership arithmetic: the ranks' rows partition the index, block by block
    for world in (1, 2, 3, 8):
        rows = np.concatenate([gdist.tri_owned_rows(1300, r, world) for r in range(world)])
        assert np.array_equal(np.sort(rows), np.arange(1300))
