"""Multi-GPU BPR: one process per GPU, users sharded by contiguous row range, item factors replicated.

The reference trains in one process; sharding is this implementation's own
(SURVEY.md 8e).  Per epoch every rank runs its share of the SGD samples against its own replica of Q
and the ranks then exchange the ONE thing they share:

    Q  <-  Q_sync + sum over ranks of (Q_rank - Q_sync)          (all-reduce of I*d fp32 over RCCL)

`run_epoch` is backend-agnostic: `engine` is anything with epoch()/export_delta()/import_delta()
(gorse_amd.dist.HipEngine on a GPU; the CPU tests inject an oracle-backed engine and run the very
same function over gloo with world_size 2).
"""
import numpy as np


def shard_range(n_rows, rank, world, align=1):
    """Contiguous row range [lo, hi) owned by `rank` (remainder rows go to the first ranks).  With `align` the inner
    boundaries are rounded to the NEAREST multiple of it (the symmetric form of the dense all-pairs sweep needs the first
    query row on a tile boundary: gorse_amd/csrc/topk_mfma.hip, 128 rows) -- but only where a shard is much larger than the
    alignment (n_rows / world >= 8 * align): rounding the boundaries of small shards empties some and doubles others (200 rows
    over 4 ranks at 128: (0,0) (0,0) (0,128) (128,200)), and a search of that size takes the scan anyway, which needs no alignment."""
    base, rem = divmod(int(n_rows), int(world))
    aligned = align > 1 and base >= 8 * align

    def edge(r):
        at = r * base + min(r, rem)
        return at if r == world or r == 0 or not aligned else (at + align // 2) // align * align
    return edge(rank), edge(rank + 1)


def samples_for_rank(n_total, users_with_feedback_local, users_with_feedback_total):
    """The reference draws the user uniformly among users with feedback (model/cf/model.go:452-458),
    so a shard's expected share of an epoch's CountFeedback() samples is its share of such users."""
    if users_with_feedback_total == 0:
        return 0
    return int(round(n_total * users_with_feedback_local / users_with_feedback_total))


def shard_csr(indptr, indices, lo, hi):
    """Rows [lo, hi) of a CSR, re-based to start at 0."""
    indptr = np.asarray(indptr, np.int64)
    p = indptr[lo:hi + 1] - indptr[lo]
    return np.ascontiguousarray(p), np.ascontiguousarray(np.asarray(indices)[indptr[lo]:indptr[hi]])


def run_epoch(engine, comm, n_samples, lr, reg, seed, epoch, sample_base):
    """One data-parallel BPR epoch + the item-factor exchange.  comm: object with all_reduce_sum(tensor)
    (None for a single rank), or a LibComm: the exchange then is ONE library call (gorse_mf_item_allreduce: export
    kernel, RCCL all-reduce and import kernel enqueued on the handle's stream, no host synchronisation)."""
    engine.epoch(n_samples, lr, reg, seed, epoch, sample_base)
    if isinstance(comm, LibComm):
        if comm.world > 1 or comm.always:
            comm.item_allreduce(engine.mf)
        return
    if comm is not None and comm.world > 1:
        delta = engine.export_delta()
        comm.all_reduce_sum(delta)
        engine.import_delta(delta)


def block_rows(n_rows, world):
    """Rows of the largest shard: the slot size of the equal-sized blocks an all-gather moves."""
    return shard_range(n_rows, 0, world)[1]


def run_als_epoch(engine, comm, weight, reg):
    """One row-sharded ALS epoch: every rank holds both factor matrices and the whole dataset,
    solves its own user rows, the ranks all-gather the user row blocks, then the same for the item rows.  The
    d x d Gram matrices are recomputed locally from the gathered (identical) replicas, so the only exchange is the
    two all-gathers: (U + I) * d fp32 per epoch.  engine: half(side), export_block(side) -> tensor of
    block_rows * d, import_blocks(side, gathered tensor of world * block_rows * d)."""
    for side in (0, 1):  # model.go:645-690, then :693-738
        engine.half(side, weight, reg)
        if isinstance(comm, LibComm):
            if comm.world > 1 or comm.always:
                comm.rows_allgather(engine.mf, side, [shard_range(engine.rows[side], r, comm.world)[0] for r in range(comm.world)]
                                    + [engine.rows[side]])
            continue
        if comm is not None and comm.world > 1:
            mine = engine.export_block(side)
            gathered = comm.all_gather(mine)
            engine.import_blocks(side, gathered)


def evaluate_sharded(engine, comm, topk, metrics=("ndcg", "precision", "recall")):
    """Evaluate (model/cf/evaluator.go:35-72) over user shards: every rank is one parallel.Parallel worker with its
    partial sums (partSum[workerId]) and partial count, the partials are added across ranks (one all-reduce of
    len(metrics) + 1 floats) and scaled by 1 / count -- the reference's own reduction, with ranks for workers.
    engine.eval_partial(topk, metrics) -> (float32 sums, float32 count) over the users this rank owns."""
    sums, count = engine.eval_partial(topk, metrics)
    t = np.array([float(x) for x in sums] + [float(count)], np.float32)
    if isinstance(comm, LibComm):
        if comm.world > 1 or comm.always:
            t = comm.comm.allreduce_f32(t)
    elif comm is not None and comm.world > 1:
        import torch
        tt = torch.from_numpy(t).to(getattr(engine, "device", "cpu"))
        comm.all_reduce_sum(tt)
        t = tt.cpu().numpy()
    return (t[:-1] * np.float32(1.0) / t[-1]).astype(np.float32) if t[-1] > 0 else np.full(len(metrics), np.nan, np.float32)


def refresh_neighbors_sharded(engine, comm, k, gather=True):
    """The item-to-item / user-to-user refresh over row shards ("all-pairs top-k": query rows split, the
    index replicated, no exchange inside the search): rank r answers the stored rows of its shard_range against the whole
    index.  With `gather` the (row ids, scores, counts) blocks are all-gathered so that every rank -- the one that writes
    the neighbour lists in particular -- holds all n_rows results; three collectives of block_rows * k (* 4 bytes) each.
    engine.n_rows, engine.neighbors(lo, hi, k) -> (idx int32 (hi-lo) x k, score float32, cnt int32)."""
    world = comm.world if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    align = getattr(engine, "shard_align", 1)
    lo, hi = shard_range(engine.n_rows, rank, world, align)
    idx, score, cnt = engine.neighbors(lo, hi, k)
    if world == 1 or not gather:
        return idx, score, cnt
    import torch
    spans = [shard_range(engine.n_rows, r, world, align) for r in range(world)]
    b = max(h - l for l, h in spans)
    dev = getattr(engine, "device", "cpu")

    def padded(a, fill, width):
        out = np.full((b, width), fill, a.dtype)
        out[:hi - lo] = a.reshape(hi - lo, width)
        return torch.from_numpy(out.ravel()).to(dev)
    g_idx = comm.all_gather(padded(idx, -1, k)).cpu().numpy().reshape(world, b, k)
    g_score = comm.all_gather(padded(score, -np.inf, k)).cpu().numpy().reshape(world, b, k)
    g_cnt = comm.all_gather(padded(cnt, 0, 1)).cpu().numpy().reshape(world, b)
    return (np.concatenate([g_idx[r, :h - l] for r, (l, h) in enumerate(spans)]),
            np.concatenate([g_score[r, :h - l] for r, (l, h) in enumerate(spans)]),
            np.concatenate([g_cnt[r, :h - l] for r, (l, h) in enumerate(spans)]))


def tri_owned_rows(n_rows, rank, world, block=512):
    """Rows (of an all-pairs search over n_rows stored rows) whose results rank `rank` of a TRIANGLE-sharded search ends up
    holding: the query blocks (512 rows: a workgroup of the symmetric sweep) rank, rank + world, ..."""
    blocks = np.arange(rank, -(-n_rows // block), world, dtype=np.int64)
    rows = (blocks[:, None] * block + np.arange(block, dtype=np.int64)[None, :]).ravel()
    return rows[rows < n_rows]


def refresh_neighbors_triangle(engine, comm, k, gather=True):
    """The dense all-pairs refresh with the TRIANGLE of the symmetric sweep sharded over the ranks (include/gorse_hip.h,
    gorse_topk_tri_*): rank r sweeps the query blocks r, r + world, ... -- their own candidate lists and what
    they find for the rows of every earlier block -- so that the ranks together do exactly the single-rank symmetric sweep, where
    refresh_neighbors_sharded (query rows split) keeps the symmetric saving on each rank's diagonal square only.  Two exchanges:
    an all-gather of the pilot thresholds (4 bytes per query) in front of the main sweep, an all-to-all of the foreign candidate
    lists (~130 entries of 8 bytes per query in all) behind it; rescoring and tie path run on a query's owner.
    engine: begin(k, rank, world), slice(r) -> (lo, hi, owned), thresholds() -> own slice, put_thresholds(lo, hi, a), sweep(),
    pack(dest) -> (counts int32, entries as int64 words), unpack(src, counts, entries), finish() -> (idx, dist) with this rank's
    rows filled in.  Thresholds and messages are numpy arrays or torch tensors (HipTriEngine(device="cuda"): they never leave the GPU).
    comm: all_gather_var(array) -> list per rank, all_to_all_var(list per dest) -> list per source (TorchComm), or None.
    Returns (idx, dist) of all n_rows rows when gather (every rank), else this rank's arrays (foreign rows -1 / +inf)."""
    world = comm.world if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    engine.begin(k, rank, world)
    if world > 1:
        parts = comm.all_gather_var(engine.thresholds())
        for r, a in enumerate(parts):
            if r != rank:
                lo, hi, _ = engine.slice(r)
                engine.put_thresholds(lo, hi, a)
    engine.sweep()
    if world > 1:
        out = [engine.pack(d) if d != rank else None for d in range(world)]
        like = next(o for o in out if o is not None)
        out[rank] = (like[0][:0], like[1][:0])
        counts_in = comm.all_to_all_var([c for c, _ in out])    # int32 per owned query
        entries_in = comm.all_to_all_var([e for _, e in out])   # (key, row) pairs as 64-bit words (int64: what torch moves)
        for s_ in range(world):
            if s_ != rank:
                engine.unpack(s_, counts_in[s_], entries_in[s_])
    idx, dist = engine.finish()
    if world == 1 or not gather or idx is None:
        return idx, dist
    own = tri_owned_rows(engine.n_rows, rank, world)
    g_idx = comm.all_gather_var(np.ascontiguousarray(idx[own]).ravel())
    g_dist = comm.all_gather_var(np.ascontiguousarray(dist[own]).ravel())
    for r in range(world):
        rows = tri_owned_rows(engine.n_rows, r, world)
        idx[rows] = g_idx[r].reshape(rows.size, k)
        dist[rows] = g_dist[r].reshape(rows.size, k)
    return idx, dist


def refresh_neighbors_triangle_local(engines, k, timings=None):
    """The same search with all `world` ranks in THIS process, one engine (handle) per rank -- N devices of one node driven by one
    process (the Go master's mode), or N handles on ONE device: the emulation the single-GPU box can run, every exchange through
    host memory.  The protocol is refresh_neighbors_triangle's, stage by stage over all ranks.  timings (a dict): per rank the
    seconds of its stages [pilots, sweep, pack (kernels), unpack (host -> device copy + kernel), finish, message device -> host],
    measured around calls that end synchronised; the last one and the copy inside unpack stand in for the transfers over xGMI."""
    import time
    world = len(engines)
    tm = [[0.0] * 6 for _ in range(world)]

    def timed(r, stage, fn):
        t0 = time.perf_counter()
        out = fn()
        engines[r].synchronize()
        tm[r][stage] += time.perf_counter() - t0
        return out
    for r, e in enumerate(engines):
        timed(r, 0, lambda: e.begin(k, r, world))
    thr = [e.thresholds() for e in engines]
    for r, e in enumerate(engines):
        for s_ in range(world):
            if s_ != r:
                lo, hi, _ = e.slice(s_)
                e.put_thresholds(lo, hi, thr[s_])
    for r, e in enumerate(engines):
        timed(r, 1, e.sweep)
    msgs = {}
    for r, e in enumerate(engines):
        for d in range(world):
            if d != r:
                if hasattr(e, "pack_device"):
                    sizes = timed(r, 2, lambda: e.pack_device(d))
                    msgs[(r, d)] = timed(r, 5, lambda: e.pack_fetch(*sizes))
                else:
                    msgs[(r, d)] = timed(r, 2, lambda: e.pack(d))
    for (src, d), (c, ent) in msgs.items():
        timed(d, 3, lambda: engines[d].unpack(src, c, ent))
    idx = dist = None
    for r, e in enumerate(engines):
        idx, dist = timed(r, 4, lambda: e.finish(idx, dist))  # every rank writes its own rows into the same arrays
    if timings is not None:
        timings["per_rank_seconds"] = tm
        timings["message_bytes"] = {"%d->%d" % kd: int(c.nbytes + ent.nbytes) for kd, (c, ent) in msgs.items()}
    return idx, dist


class HipTriEngine:
    """A gorse_topk handle (capi.TopK) as the engine of refresh_neighbors_triangle.  device = "cpu": thresholds and messages as host
    numpy arrays (the one-GPU emulation, gloo); device = "cuda": as torch CUDA tensors filled and read by the library through their
    device pointers -- under RCCL the foreign lists go GPU to GPU."""

    def __init__(self, index, fetch=True, device="cpu"):
        self.index, self.n_rows, self.fetch, self.device = index, index.N, fetch, device
        self.rank = 0
        self.torch = None
        if device != "cpu":
            import torch
            self.torch = torch

    def begin(self, k, rank, world):
        self.rank = rank
        self.index.tri_begin(k, rank, world)

    def slice(self, r):
        return self.index.tri_slice(r)

    def thresholds(self):
        lo, hi, _ = self.index.tri_slice(self.rank)
        if self.torch is None:
            return self.index.tri_thresholds_get(lo, hi)
        t = self.torch.empty(hi - lo, dtype=self.torch.float32, device=self.device)
        self.index.tri_thresholds_get_ptr(lo, hi, t.data_ptr())
        return t

    def put_thresholds(self, lo, hi, a):
        if self.torch is None or not self.torch.is_tensor(a):
            return self.index.tri_thresholds_put(lo, hi, a)
        a = a.to(self.device).contiguous()
        self.torch.cuda.synchronize()  # the collective that filled it ran on torch's stream
        self.index.tri_thresholds_put_ptr(lo, hi, a.data_ptr())

    def sweep(self):
        self.index.tri_sweep()

    def pack(self, dest):
        if self.torch is None:
            c, e = self.index.tri_pack(dest)
            return c, e.view(np.int64)
        nc, ne = self.index.tri_pack_device(dest)
        c = self.torch.empty(nc, dtype=self.torch.int32, device=self.device)
        e = self.torch.empty(ne, dtype=self.torch.int64, device=self.device)
        self.index.tri_pack_read_ptr(c.data_ptr(), e.data_ptr())
        return c, e

    def pack_device(self, dest):
        return self.index.tri_pack_device(dest)

    def pack_fetch(self, nc, ne):
        c, e = self.index.tri_pack_fetch(nc, ne)
        return c, e.view(np.int64)

    def unpack(self, src, counts, entries):
        if self.torch is None or not self.torch.is_tensor(counts):
            return self.index.tri_unpack(src, counts, np.asarray(entries).view(np.uint64))
        counts, entries = counts.to(self.device).contiguous(), entries.to(self.device).contiguous()
        self.torch.cuda.synchronize()
        self.index.tri_unpack_ptr(src, counts.data_ptr(), counts.numel(), entries.data_ptr(), entries.numel())

    def finish(self, idx=None, dist=None):
        return self.index.tri_finish(idx, dist, fetch=self.fetch)

    def synchronize(self):
        self.index.synchronize()


class HipNeighborsEngine:
    """A gorse_sparse (capi.Sparse) or gorse_topk (capi.TopK) handle as the engine of refresh_neighbors_sharded."""

    def __init__(self, index, device="cuda", fetch=True):
        self.index, self.device, self.n_rows, self.fetch = index, device, index.N, fetch
        # a dense index: shards start on a tile boundary, so that every rank's pass can take the symmetric form of the sweep
        self.shard_align = 128 if hasattr(index, "last_symmetric") else 1

    def neighbors(self, lo, hi, k):
        if not self.fetch:  # results stay in HBM (bench.py's timed region); nothing to gather
            self.index.all_pairs(k, lo, hi, fetch=False)
            return None, None, None
        res = self.index.all_pairs(k, lo, hi)
        if len(res) == 3:  # Sparse: (idx, score, cnt)
            return res
        idx, dist = res    # TopK: distances ascending, padded with -1 / +inf
        return idx, dist, (idx >= 0).sum(axis=1).astype(np.int32)


class LibComm:
    """The library's own RCCL communicator (gorse_comm_*, csrc/comm.hip) for one process per GPU: rank 0 draws the unique
    id, `share(bytes) -> bytes` ships it to every rank (bench.py: a torch.distributed broadcast).  `always` runs the
    collectives at world 1 too (what the one-GPU box can test of this path)."""

    def __init__(self, rank, world, device, share, always=False, agree=None):
        """agree(ok: bool) -> bool: true iff every rank passed true (bench.py: an all-reduce MIN).  gorse_comm_create is a
        collective initialisation, so a rank that cannot open RCCL must say so BEFORE the others enter it."""
        from . import capi
        self.capi = capi
        why = capi.comm_available()
        if agree is not None and not agree(why is None):
            raise RuntimeError("RCCL is unavailable on some rank" + (": " + why if why else ""))
        if agree is None and why is not None:
            raise RuntimeError("RCCL unavailable: " + why)
        # Every rank reaches share() whatever happens on rank 0 (an id of zeros = "rank 0 has none"): a rank that raised before
        # the exchange would leave the others waiting in it
        none, err = bytes(capi.COMM_ID_BYTES), None
        uid = none
        if rank == 0:
            try:
                uid = capi.Comm.unique_id()
            except Exception as e:  # e.g. librccl.so cannot be loaded
                err = e
        uid = share(uid)
        if uid == none:
            raise err or RuntimeError("rank 0 could not draw an RCCL unique id")
        self.comm = capi.Comm(uid, world, rank, device)
        self.world, self.rank, self.always = world, rank, always

    def item_allreduce(self, mf):
        self.capi.item_allreduce([mf], [self.comm])

    def rows_allgather(self, mf, side, row_splits):
        self.capi.rows_allgather([mf], [self.comm], side, row_splits)

    def close(self):
        self.comm.close()


class LocalComms:
    """One process, N GPUs -- the Go master's mode (integration/go/model/cf/rccl_hip.go): gorse_comm_create_local gives one
    communicator per device, and every exchange is ONE library call that receives all N (handle, communicator) pairs and
    issues their collectives as one RCCL group."""

    def __init__(self, devices):
        from . import capi
        self.capi = capi
        self.comms = capi.Comm.local(devices)
        self.world = len(self.comms)

    def item_allreduce(self, mfs):
        self.capi.item_allreduce(mfs, self.comms)

    def rows_allgather(self, mfs, side, row_splits):
        self.capi.rows_allgather(mfs, self.comms, side, row_splits)

    def close(self):
        for c in self.comms:
            c.close()


def run_epoch_local(engines, comms, samples, lr, reg, seed, epoch):
    """hipGroup.bprEpoch of rccl_hip.go: the epoch of EVERY device is enqueued first (nothing waits for a device), then the
    item factors are summed by one grouped all-reduce over all (handle, communicator) pairs.  engines[r] drives the handle
    of shard r (its sample stream starts at r << 40), samples[r] = its share of the epoch's samples."""
    for r, (eng, n) in enumerate(zip(engines, samples)):
        eng.epoch(n, lr, reg, seed, epoch, r * (1 << 40))
    comms.item_allreduce([e.mf for e in engines])


def run_als_epoch_local(engines, comms, weight, reg):
    """alsEpochSharded of rccl_hip.go: per half-sweep every device's kernels are enqueued (gorse_als_half_epoch_enqueue -- the
    synchronous call would make the devices take turns), then one grouped all-gather of the row blocks; one host
    synchronisation per epoch."""
    world = len(engines)
    for side in (0, 1):
        for eng in engines:
            eng.mf.als_half_epoch_enqueue(side, weight, reg)
        rows = engines[0].rows[side]
        comms.rows_allgather([e.mf for e in engines], side, [shard_range(rows, r, world)[0] for r in range(world)] + [rows])
    for eng in engines:
        eng.mf.synchronize()


class TorchComm:
    """torch.distributed plumbing (backend 'nccl' = RCCL over xGMI on ROCm, 'gloo' in the CPU tests)."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0

    def _staged(self, tensor):
        """gloo moves host memory: a device tensor goes through a host copy (the functional multi-rank check of bench.py on a box
        with fewer GPUs than ranks; under nccl = RCCL nothing is staged)"""
        return self.dist.is_initialized() and self.dist.get_backend() == "gloo" and tensor.is_cuda

    def all_reduce_sum(self, tensor):
        if self._staged(tensor):
            host = tensor.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM)
            tensor.copy_(host)
            return
        self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM)

    def all_gather(self, tensor):
        """concatenation over ranks of equal-sized 1-D tensors"""
        import torch
        if self._staged(tensor):
            parts = [torch.empty(tensor.numel(), dtype=tensor.dtype) for _ in range(self.world)]
            self.dist.all_gather(parts, tensor.cpu())
            return torch.cat(parts).to(tensor.device)
        if self.dist.is_initialized() and self.dist.get_backend() == "gloo":
            parts = [torch.empty_like(tensor) for _ in range(self.world)]
            self.dist.all_gather(parts, tensor)
            return torch.cat(parts)
        out = torch.empty(self.world * tensor.numel(), dtype=tensor.dtype, device=tensor.device)
        self.dist.all_gather_into_tensor(out, tensor)
        return out

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def _xdev(self):
        """where a message of host arrays travels: the GPU under nccl (= RCCL), host memory under gloo"""
        import torch
        return torch.device("cuda", torch.cuda.current_device()) if self.dist.get_backend() == "nccl" else torch.device("cpu")

    def all_gather_var(self, a):
        """all-gather of 1-D arrays of one dtype whose lengths differ by rank -> the list of every rank's array.  numpy in, numpy
        out; a torch tensor in (a CUDA tensor under RCCL: nothing touches the host), torch tensors out."""
        import torch
        as_torch = torch.is_tensor(a)
        if self.world == 1:
            return [a]
        dev = self._xdev()
        t = (a if as_torch else torch.from_numpy(np.ascontiguousarray(a))).to(dev)
        n = torch.tensor([t.numel()], dtype=torch.int64, device=dev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
        self.dist.all_gather(sizes, n)
        sizes = [int(x.item()) for x in sizes]
        mine = torch.zeros(max(max(sizes), 1), dtype=t.dtype, device=dev)
        mine[:t.numel()] = t
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        parts = [p[:sizes[r]] for r, p in enumerate(parts)]
        return [p.to(a.device) for p in parts] if as_torch else [p.cpu().numpy() for p in parts]

    def all_to_all_var(self, out):
        """out[d] = the 1-D array (one dtype for all; numpy or torch) this rank sends to rank d -> the list of what every rank sent
        here.  Sizes first (an all-gather of the size row), then one point-to-point pair per (source, destination) -- send / recv
        exist under both RCCL and gloo; under RCCL the payloads stay in device memory."""
        import torch
        as_torch = torch.is_tensor(out[0])
        if self.world == 1:
            return [out[0]]
        dev = self._xdev()
        send = [(o if as_torch else torch.from_numpy(np.ascontiguousarray(o))).to(dev).contiguous() for o in out]
        row = torch.tensor([o.numel() for o in send], dtype=torch.int64, device=dev)
        rows = [torch.zeros_like(row) for _ in range(self.world)]
        self.dist.all_gather(rows, row)
        sizes_in = [int(rows[s_][self.rank].item()) for s_ in range(self.world)]
        recv = [torch.empty(sizes_in[s_], dtype=send[0].dtype, device=dev) for s_ in range(self.world)]
        ops = []
        for peer in range(self.world):
            if peer == self.rank:
                continue
            if send[peer].numel():
                ops.append(self.dist.P2POp(self.dist.isend, send[peer], peer))
            if sizes_in[peer]:
                ops.append(self.dist.P2POp(self.dist.irecv, recv[peer], peer))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        recv[self.rank] = send[self.rank]
        return [r_.to(out[0].device) for r_ in recv] if as_torch else [r_.cpu().numpy() for r_ in recv]


class HipEngine:
    """A gorse_mf handle + a torch CUDA buffer that carries the item-factor delta through RCCL."""

    def __init__(self, mf, mode, device="cuda"):
        self.mf, self.mode = mf, mode
        self.xbuf = None
        self.torch = None  # imported only when a collective is needed (single-GPU use never touches torch)
        self.device = device  # where the exchange buffers live ("cpu" only in the gloo tests, with a stand-in handle)

    def _sync(self):
        if self.device != "cpu":
            self.torch.cuda.synchronize()

    def enable_exchange(self):
        import torch
        self.torch = torch
        self.xbuf = self.torch.empty(self.mf.I * self.mf.d, dtype=self.torch.float32, device=self.device)
        self.mf.item_sync_mark()
        self.mf.synchronize()

    def epoch(self, n_samples, lr, reg, seed, epoch, sample_base):
        self.mf.bpr_epoch_enqueue(n_samples, lr, reg, seed, epoch, sample_base=sample_base, mode=self.mode)

    def export_delta(self):
        self.mf.item_delta_export(self.xbuf.data_ptr())  # synchronises the library's stream
        return self.xbuf

    def set_eval(self, test_ptr, test_idx, neg_ptr, neg_idx):
        """test split of THIS rank's users (local user ids): positives + sampled negatives per user"""
        self.eval_split = (np.asarray(test_ptr, np.int64), np.asarray(test_idx, np.int32),
                           np.asarray(neg_ptr, np.int64), np.asarray(neg_idx, np.int32))

    def eval_partial(self, topk, metrics):
        from . import metrics as M
        test_ptr, test_idx, neg_ptr, neg_idx = self.eval_split
        users = np.nonzero(np.diff(test_ptr) > 0)[0].astype(np.int32)
        cptr, cidx = M.candidates(test_ptr, test_idx, neg_ptr, neg_idx, users)
        rank, rlen = self.mf.rank(users, cptr, cidx, topk)  # Rank + heap.TopKFilter on the device
        return M.partial_sums(rank, rlen, users, test_ptr, test_idx, metrics)

    def import_delta(self, delta):
        self._sync()  # the collective ran on torch's stream
        self.mf.item_delta_import(delta.data_ptr())


class HipAlsEngine:
    """A gorse_mf handle restricted to this rank's row ranges + torch CUDA buffers for the all-gathers."""

    def __init__(self, mf, rank, world, device="cuda", staging=True):
        """staging = False: no torch buffers (the exchange runs inside the library: LibComm / gorse_mf_rows_allgather)"""
        self.torch = None
        self.device = device
        self.mf, self.rank, self.world = mf, rank, world
        self.rows = (mf.U, mf.I)
        self.range = [shard_range(n, rank, world) for n in self.rows]
        mf.als_set_ranges(self.range[0][0], self.range[0][1], self.range[1][0], self.range[1][1])
        self.block = [block_rows(n, world) for n in self.rows]
        self.buf = None
        if world > 1 and staging:
            import torch
            self.torch = torch
            self.buf = [torch.zeros(b * mf.d, dtype=torch.float32, device=device) for b in self.block]

    def half(self, side, weight, reg):
        if self.buf is None:  # no torch staging: a single GPU, or the exchange inside the library (stream-ordered): enqueue only
            self.mf.als_half_epoch_enqueue(side, weight, reg)
        else:
            self.mf.als_half_epoch(side, weight, reg)

    def export_block(self, side):
        lo, hi = self.range[side]
        self.mf.rows_export(side, lo, hi, self.buf[side].data_ptr())  # synchronises the library's stream
        return self.buf[side]

    def import_blocks(self, side, gathered):
        if self.device != "cpu":
            self.torch.cuda.synchronize()  # the collective ran on torch's stream
        d, b = self.mf.d, self.block[side]
        for r in range(self.world):
            if r == self.rank:
                continue
            lo, hi = shard_range(self.rows[side], r, self.world)
            if hi > lo:
                self.mf.rows_import(side, lo, hi, gathered.data_ptr() + r * b * d * 4)
