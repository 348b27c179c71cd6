#!/usr/bin/env python3
"""bench.py -- BPR positive-samples/s (and item x item top-k pairs/s) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU.  A "step" is one BPR epoch = CountFeedback()
SGD samples over one rank's resident dataset (model/cf/model.go:446-494).  Rank 0 prints ONE
JSON line.

Workloads (BASELINE.json configs):
  ml1m   (default) C2: S-ml1m shape 6040 x 3706 x 994,169, nFactors 64, fp32.  With N ranks every
         rank trains its own S-ml1m-shaped USER SHARD against the shared item matrix (weak scaling:
         N*6040 users), Q replicated, one all-reduce of the item-factor delta per epoch over RCCL.
  c3     C3 shard: 125,000 users x 200,000 items x 12.5M feedbacks per rank, nFactors 128 (at N = 8
         this is exactly the 1M x 200K x 100M configuration).
  c3full C3 whole on ONE GPU: 1M users x 200K items x 100M feedbacks, nFactors 128 (P = 512 MB + Q = 102 MB: outside the
         256 MiB Infinity Cache).  The default single-GPU line carries it as "c3".
  big    north_star's "10M x 1M x 128 synthetic set" on ONE GPU: 10M users x 1M items x ~1.0e9 feedbacks (this repo's
         choice: 100 per user as in C3), nFactors 128 (P = 5.1 GB).  Builder-run (its JSON is kept under profiles/).
  ml100k C1 shape (943 x 1682 x 99,057), nFactors 16 -- the reference's own default width (model_test.go:35-45); the
         default line carries it as "ml100k" and, with nFactors 8, as "ml100k_d8".
  topk   C4 alone: item x item cosine top-100 over 1M x 128 bf16 (the default run appends it as "topk").
  als    C5: eALS 500K x 100K x 50M feedbacks, nFactors 64; rows sharded over the ranks (strong scaling), two
         all-gathers of factor row blocks per epoch.
  i2i    SURVEY 8f item 2: the "users" item-to-item refresh = sparse all-pairs top-100 over the IDF vectors of the C3-shard
         dataset's 200,000 items (item -> its users, value sqrt(idf)); query rows sharded over the ranks, index replicated,
         no collective.  `--i2i-shape ml1m` uses the S-ml1m items instead.
         The default single-GPU run appends this leg as "i2i", measured in a child process with a time limit.
Timed region: inputs resident in HBM, barrier + device sync on both sides, MAX over ranks.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

# Every handle runs two streams side by side (update / preparation of the next chunk; list walk / heavy-query kernel), and the HIP
# runtime multiplexes a process's streams onto FOUR hardware queues per device unless told otherwise; a rank of an N > 1 run also
# holds torch's and RCCL's streams, and two streams on one queue run one after the other ("Streams and hardware
# queues"; profiles/r05_zzc: two ranks sharing one GPU 55.5 -> 50.6 ms per step with eight, the single-process legs unchanged).
# Must be in the environment before the runtime initialises: hence here, before anything imports it.  The caller's value wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: libgorse_hip then binds to the HIP runtime torch already loaded)
import torch.distributed as dist  # noqa: E402

from gorse_amd import capi, synth  # noqa: E402
from gorse_amd import dist as gdist  # noqa: E402

# Where the few scalars of the run's own bookkeeping (max-over-ranks of a time, agreement flags) live for a collective: the GPU under
# nccl (= RCCL, what the driver runs); host memory under GORSE_BENCH_BACKEND=gloo -- a FUNCTIONAL check of the N > 1 code path on a box
# with fewer GPUs than ranks (all ranks then share the visible devices round robin and the exchange goes through host staging:
# scripts/gpu_session.sh stage `ranks2`; its numbers mean nothing).
BACKEND = os.environ.get("GORSE_BENCH_BACKEND", "nccl")
COLL_DEV = "cuda" if BACKEND == "nccl" else "cpu"

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense, 2495 TF measured)
MFMA_F32_PEAK_TFLOPS = 157.3  # fp32-input MFMA peak = the fp32 vector rate (MI355X_MICROARCH.md: 157.3 TF spec, 155 measured)


TRAFFIC_ROUND = "r06"  # a PMC record is reported only if its session is of THIS round (the kernels it measured are the ones timed)


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json, written
    by scripts/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this very command).
    bench.py cannot collect counters on itself; it reports the recorded figure and says where it came from -- and REFUSES
    (traffic: null) a record whose session is not of the current round: a figure measured on an earlier round's kernels says
    nothing about the ones this run timed."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[workload]
        f, w = rec.get("fetch_bytes_per_launch"), rec.get("write_bytes_per_launch")
        if f is None or w is None:
            return None, None
        if not str(rec.get("session", "")).startswith(TRAFFIC_ROUND):
            return None, "profiles/traffic.json[%s] is a record of session %r, not of round %s: not reported" % (
                workload, rec.get("session", "?"), TRAFFIC_ROUND)
        return f + w, "profiles/traffic.json[%s]: FETCH_SIZE %.0f MB + WRITE_SIZE %.0f MB per launch; %s" % (
            workload, f / 1e6, w / 1e6, "session %s; %s" % (rec.get("session", "?"), rec.get("how", "")))  # prose: compact() moves it to the notes file
    except Exception:
        return None, None


# The driver keeps the last 8 KB of stdout: the JSON line carries numbers; everything that is prose (where a traffic figure
# came from, what a CPU sample was, how a flop count is defined) goes to a notes file keyed by the same path.
NOTE_KEYS = {"traffic_source", "sample", "flop_note", "note", "entry_point", "parallelism", "how"}


def compact(obj, notes, path=""):
    if isinstance(obj, dict):
        out = {}
        for k, v in obj.items():
            here = "%s.%s" % (path, k) if path else k
            if k in NOTE_KEYS and isinstance(v, str):
                notes[here] = v
            else:
                out[k] = compact(v, notes, here)
        return out
    if isinstance(obj, float):
        return float("%.5g" % obj)
    return obj


def write_notes(notes, tag="default"):
    for d in (os.path.join(ROOT, "gpurun_out"), "/tmp"):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, "bench_notes_%s.json" % tag)
            json.dump(notes, open(path, "w"), indent=1, sort_keys=True)
            return path
        except Exception:
            continue
    return None


class ClockSampler:
    """Shader clock of the device while a leg runs: a thread reads the current level of pp_dpm_sclk (amdgpu sysfs) every 5 ms.
    The MFMA-bound sweep follows the clock the box sustains under load (MI355X_MICROARCH.md, DVFS); its roofline fraction is
    quoted against the spec peak, so the line says what clock the number was measured at.  A box shows the sysfs nodes of ALL
    its GPUs whichever one the process may use: every card is sampled and the busiest one (highest mean clock) is reported.
    None when no such file is readable."""

    def __init__(self, local):
        self.paths, self.vals, self.stop, self.th = [], [], False, None
        try:
            for c in sorted(os.listdir("/sys/class/drm")):
                f = "/sys/class/drm/%s/device/pp_dpm_sclk" % c
                if c.startswith("card") and c[4:].isdigit() and os.path.exists(f):
                    self.paths.append(f)
        except Exception:
            self.paths = []
        self.vals = [[] for _ in self.paths]

    @staticmethod
    def _read(path):
        try:
            for line in open(path):
                if "*" in line:
                    return float(line.split(":")[1].strip().split("M")[0])
        except Exception:
            return None
        return None

    def __enter__(self):
        if self.paths:
            def loop():
                while not self.stop:
                    for k, f in enumerate(self.paths):
                        v = self._read(f)
                        if v:
                            self.vals[k].append(v)
                    time.sleep(0.005)
            self.th = threading.Thread(target=loop, daemon=True)
            self.th.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        if self.th:
            self.th.join()

    def ghz(self):
        means = [float(np.mean(v)) for v in self.vals if v]
        return max(means) / 1e3 if means else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["ml1m", "c3", "c3full", "big", "ml100k", "topk", "als", "i2i"],
                    help="default: ml1m (C2) on one GPU -- the line then also carries topk (C4), c3 (whole C3), i2i, als and "
                         "ml100k objects; the C3 shard per rank + sharded topk on N > 1")
    ap.add_argument("--factors", type=int, default=0, help="nFactors override of a BPR workload")
    ap.add_argument("--big-draws", type=int, default=1_250_000_000, help="feedback draws of --workload big (about 80 %% remain distinct)")
    ap.add_argument("--bpr-chunk", type=int, default=0, help="probe: samples per chunk of the BPR pipeline (0 = the library's choice)")
    ap.add_argument("--sync-steps", type=int, default=3, help="epochs timed through the synchronous gorse_bpr_epoch (0 = skip)")
    ap.add_argument("--comm", default="lib", choices=["lib", "torch"], help="who owns the RCCL communicator of a multi-rank run")
    ap.add_argument("--no-extra", action="store_true", help="default single-GPU run without the c3 / i2i / als objects")
    ap.add_argument("--i2i-shape", default="c3", choices=["c3", "ml1m", "ml100k"])
    ap.add_argument("--als-scale", type=float, default=1.0, help="shrink S-als (users, items, feedbacks) by this factor")
    ap.add_argument("--no-topk", action="store_true", help="skip the item x item top-k leg of the default run")
    ap.add_argument("--topk-n", type=int, default=1_000_000)
    ap.add_argument("--topk-steps", type=int, default=2)
    ap.add_argument("--topk-budget", type=float, default=60.0, help="seconds the timed top-k steps may take (see bench_topk)")
    ap.add_argument("--topk-watchdog", type=float, default=240.0, help="N > 1: seconds the sharded top-k leg may take before the line is emitted without it")
    ap.add_argument("--topk-shard", default="auto", choices=["auto", "rows", "tri"],
                    help="how the C4 pass is split over N > 1 ranks: tri = the TRIANGLE of the symmetric sweep (rank r takes the query blocks "
                         "r, r + N, ...; thresholds all-gathered, foreign candidate lists all-to-all: gorse_topk_tri_*), rows = contiguous "
                         "query-row shards (no exchange; the symmetric saving on each rank's diagonal square only); auto = tri "
                         "(DESIGN.md section 5: measured per-rank passes 122 / 71 / 46 ms at 2 / 4 / 8 ranks against 130 / 86 / 57)")
    ap.add_argument("--mode", type=int, default=capi.BPR_HOGWILD_STORES,
                    help="BPR schedule: 3 = what Fit runs with Jobs > 1 (atomics + the reference's unlocked store for cold negatives), "
                         "0 = atomics only, 1 = sequential, 2 = racy")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


BIG_DRAWS_DEFAULT_LINE = 250_000_000  # scripts/gen_golden_ndcg.py big and the -m gpu test at that size use the same set

BPR_SHAPES = {"ml1m": (64, "S-ml1m 6040x3706x994169 per rank (C2)"),
              "ml100k": (16, "S-ml100k 943x1682x99057 per rank (C1)"),
              "c3": (128, "S-big shard 125000x200000x12.5M per rank (C3/8)"),
              "c3full": (128, "S-big whole 1000000x200000x100M on one GPU (C3)"),
              "big": (128, "S-huge 10M users x 1M items (north_star's 10M x 1M x 128 set)")}


def make_data_desc(workload, factors=0):
    d, desc = BPR_SHAPES[workload]
    d = factors or d
    return d, "%s, nFactors=%d" % (desc, d)


def make_data(workload, rank, args):
    d, desc = make_data_desc(workload, args.factors)
    if workload == "ml1m":
        data = synth.synth_cf(6040, 3706, 994169, seed=42 + 1000 * rank, min_len=19, n_neg=99, with_test=False)
    elif workload == "ml100k":
        data = synth.synth_cf(943, 1682, 99057, seed=42 + 1000 * rank, min_len=19, n_neg=99, with_test=False)
    elif workload == "c3full":
        data = synth.s_big_full()
    elif workload == "big":
        data = synth.s_huge(N=args.big_draws)
    else:
        data = synth.s_big_shard(rank=rank, world=8)
    return data, d, desc


def shard_of(full, rank, world):
    """rank's user shard of a whole data set as its own CFData with both CSR sides (s_big_full lays the shards end to end)"""
    lo, hi = rank * full.U // world, (rank + 1) * full.U // world
    uptr, uidx = gdist.shard_csr(full.uptr, full.uidx, lo, hi)
    rows = np.repeat(np.arange(hi - lo, dtype=np.int64), np.diff(uptr))
    iptr, iidx = synth._csr_from_pairs(uidx.astype(np.int64), rows, full.I)
    z, e = np.zeros(hi - lo + 1, np.int64), np.zeros(0, np.int32)
    return synth.CFData(hi - lo, full.I, uptr, uidx, iptr, iidx, z, e, z.copy(), e.copy())


def cpu_baseline(data, d, lr, reg, seconds):
    """The oracle's BPR epoch (kind 'port': our C restatement of model.go:446-494, auto-vectorised
    build) timed on this host: Hogwild over T threads sharing P, Q -- like parallel.Parallel with
    Jobs = T but without the per-sample channel hop, so an upper bound for the Go path."""
    from oracle import oracle as orc
    orc.build()
    fast = os.path.join("/tmp", "liboracle_fast_%d.so" % os.getuid())
    src = os.path.join(ROOT, "oracle", "gorse_oracle.c")
    if not os.path.exists(fast) or os.path.getmtime(fast) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=gnu11", "-fPIC", "-shared", "-o", fast, src, "-lm"])
    L = ctypes.CDLL(fast)
    f32p, i32p, i64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
    L.orc_bpr_epoch_sampled.restype = ctypes.c_double
    L.orc_bpr_epoch_sampled.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, i64p, i32p, i32p,
                                        ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_float, ctypes.c_float]
    P, Q = (synth.init_factors_big if data.U * d >= 1 << 26 else synth.init_factors)(data.U, data.I, d, 0.0, 0.001, 1)
    srt = orc.sort_rows(data.uptr, data.uidx)
    uptr = np.ascontiguousarray(data.uptr, np.int64)
    uidx = np.ascontiguousarray(data.uidx, np.int32)
    threads = min(os.cpu_count() or 1, 32)

    def run(n_per_thread, epoch):
        def work(t):
            L.orc_bpr_epoch_sampled(P.ctypes.data_as(f32p), Q.ctypes.data_as(f32p), data.U, data.I, d,
                                    uptr.ctypes.data_as(i64p), uidx.ctypes.data_as(i32p), srt.ctypes.data_as(i32p), 1,
                                    epoch, t * n_per_thread, n_per_thread, lr, reg)
        ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return time.perf_counter() - t0

    probe = 50_000
    dt = run(probe, 0)
    rate = probe * threads / dt
    n = int(max(probe, min(rate * seconds / threads, 50_000_000)))
    dt = run(n, 1)
    return {"value": n * threads / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": "%d samples (%d per thread x %d Hogwild threads) of the same workload, %.1f s; "
                      "oracle/gorse_oracle.c built -O3 -march=native" % (n * threads, n, threads, dt)}


def topk_cpu_baseline(Xe, k, seconds, idx_gpu, dist_gpu, q_begin):
    """ann.Bruteforce.SearchIndex restated (oracle, kind 'port') on a few query rows, one query per host thread;
    the rows double as a bit-exact check of the GPU result."""
    from oracle import oracle as orc
    o = orc.Oracle()
    o.set_isa(orc.ISA_AVX512)
    N = Xe.shape[0]
    threads = min(os.cpu_count() or 1, 32)
    t0 = time.perf_counter()
    o.search_index(Xe, orc.METRIC_COSINE, q_begin, k)
    one = max(time.perf_counter() - t0, 1e-3)
    per_thread = max(1, min(int(seconds / one), 64))
    qs = [q_begin + 97 * t for t in range(threads * per_thread)]
    res = [None] * len(qs)

    def work(t):
        for r in range(t, len(qs), threads):
            res[r] = o.search_index(Xe, orc.METRIC_COSINE, qs[r], k)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    ok = 0
    for q, (ei, ed) in zip(qs, res):
        r = q - q_begin
        if r < idx_gpu.shape[0]:
            same = np.array_equal(idx_gpu[r, :ei.size], ei) and np.array_equal(dist_gpu[r, :ei.size].view(np.uint32), ed.view(np.uint32))
            assert same, "GPU top-k row %d differs from the oracle" % q
            ok += 1
    return {"value": len(qs) * (N - 1) / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": "%d full queries (each against all %d vectors, d=%d) on %d threads, %.1f s; %d of them compared "
                      "bit for bit with the GPU rows" % (len(qs), N, Xe.shape[1], threads, dt, ok)}


def bench_topk(args, world, rank, local, fence):
    """BASELINE config C4: item-to-item cosine top-100 over N x 128 bf16 embeddings, X replicated.  Over N > 1 ranks either the
    TRIANGLE of the symmetric sweep is sharded (--topk-shard tri, the default: rank r sweeps the query blocks r, r + N,
    ..., the pilot thresholds are all-gathered and the foreign candidate lists exchanged all-to-all inside the timed region) or the
    query rows (rows: no collective).  A step = one all-pairs pass (this rank's share of it); the N x k indices +
    distances stay in HBM (timed region: embeddings resident -> results resident)."""
    N, d, k = args.topk_n, 128, 100
    Xb, Xe = synth.s_emb(N, d, 44)
    t = capi.TopK(Xb, capi.METRIC_COSINE, dtype=capi.DTYPE_BF16, device=local)
    tri = world > 1 and N >= 1 << 17 and args.topk_shard in ("tri", "auto")
    if tri:
        # the triangle of the symmetric sweep sharded over the ranks: every rank's step covers ALL N query rows' share of the work
        # (its query blocks r, r + world, ...), the exchanges (thresholds, foreign lists) are inside the timed region
        eng = gdist.HipTriEngine(t, fetch=False, device="cuda" if BACKEND == "nccl" else "cpu")
        tcomm = gdist.TorchComm()
        gdist.refresh_neighbors_triangle(eng, tcomm, k, gather=False)  # allocations, code objects
        gdist.refresh_neighbors_triangle(eng, tcomm, k, gather=False)
        fence()
        t.set_profiling(True)
        with ClockSampler(local) as clk:
            t0 = time.perf_counter()
            for _ in range(args.topk_steps):
                gdist.refresh_neighbors_triangle(eng, tcomm, k, gather=False)
            fence()
            dt = time.perf_counter() - t0
        q0, q1 = 0, N
        full = N
        own_queries = t.tri_slice(rank)[2]
    else:
        q0, q1 = gdist.shard_range(N, rank, world, align=128)  # tile-aligned shards: every rank's pass can take the symmetric sweep
        t.all_pairs(k, q0, min(q1, q0 + 8192), fetch=False)  # warm-up: allocations, code objects
        fence()
        # keep the default run bounded: if a 16K-query pass predicts more than --topk-budget seconds for the timed
        # steps, time a prefix of the query rows instead (all N stored vectors are still scanned per query; the
        # pairs/s figure is over the rows actually processed and config.queries_per_step_per_gpu says how many)
        probe_q = min(q1 - q0, 16384)
        t0 = time.perf_counter()
        t.all_pairs(k, q0, q0 + probe_q, fetch=False)
        fence()
        per_query = (time.perf_counter() - t0) / probe_q
        full = q1 - q0
        if per_query * full * args.topk_steps > args.topk_budget:
            q1 = q0 + max(probe_q, int(args.topk_budget / args.topk_steps / per_query) // 8192 * 8192)
            q1 = min(q1, q0 + full)
        # one untimed pass over exactly the timed query range: the tie path sizes its buffers by the number of queries with
        # ties (1 % of them), so a short warm-up leaves allocations inside the first timed step
        t.all_pairs(k, q0, q1, fetch=False)
        fence()
        t.set_profiling(True)
        with ClockSampler(local) as clk:
            t0 = time.perf_counter()
            for _ in range(args.topk_steps):
                t.all_pairs(k, q0, q1, fetch=False)
            fence()
            dt = time.perf_counter() - t0
        own_queries = q1 - q0
    tt = torch.tensor([dt], dtype=torch.float64, device=COLL_DEV)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    launches, sweep_ms = t.get_profile(capi.PROF_TOPK_SWEEP)
    r_launches, resc_ms = t.get_profile(capi.PROF_TOPK_SELECT)
    h_launches, hist_ms = t.get_profile(capi.PROF_TOPK_HIST)
    p_launches, replay_ms = t.get_profile(capi.PROF_TOPK_REPLAY)
    t.set_profiling(False)
    n_fb, n_tie = t.last_stats()
    symmetric, sym_stats = t.last_symmetric(), t.sym_stats()
    if rank != 0:
        return None
    # rows sharded: every rank answers its q1 - q0 query rows; triangle: the ranks TOGETHER answer all N (a rank owns own_queries of them)
    pairs_step = (q1 - q0) * (N - 1) / (world if tri else 1)
    flops_launch = 2.0 * d * (q1 - q0) / (world if tri else 1) * N * args.topk_steps / max(launches, 1)  # 2*d flop per scored pair
    avg_ms = sweep_ms / max(launches, 1)
    achieved = flops_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
    out = {
        "metric": "item x item cosine top-%d pairs/sec (whole job, N GPUs)" % k,
        "value": world * pairs_step * args.topk_steps / dt, "unit": "pairs/s",
        "n_gpus": world, "steps": args.topk_steps, "ms_per_step": dt / args.topk_steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "S-emb %dx%d bf16, cosine, k=%d (C4), %s x%d" % (N, d, k, "the triangle of the symmetric sweep sharded" if tri else "query rows sharded", world),
                   "sharding": "triangle" if tri else "rows",
                   "queries_per_step_per_gpu": own_queries, "all_query_rows": q1 - q0 == full,
                   "tie_replayed_queries": n_tie, "scan_fallback_queries": n_fb, "symmetric_sweep": symmetric,
                   "queries_without_pilot_threshold": sym_stats[0], "foreign_lists_overflowed": sym_stats[2]},
        "roofline": {"bound": "mfma", "kernel": "topk_sweep_kernel", "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": None,
                     "algorithmic_flop_per_pair": 2 * d, "avg_launch_ms": avg_ms, "launches": launches,
                     "flop_note": "achieved = ALGORITHMIC flops (2 d per scored (query, row) pair, SURVEY 8d) over the sweep launches' time; "
                                  "the symmetric form of the sweep (csrc/topk_mfma.hip, SYM) issues about half of them on the matrix cores -- "
                                  "every score of the square block [q0, q1) x [q0, q1) serves both of its queries -- so MFMA-busy counters read "
                                  "lower than this fraction" if symmetric else "achieved = algorithmic flops (2 d per scored pair) over the sweep launches' time",
                     "clock_ghz": clk.ghz(),  # mean shader clock over the timed steps (pp_dpm_sclk), None if unreadable
                     "rescore_avg_ms": resc_ms / max(r_launches, 1),
                     "tie_history_sweep_ms_per_step": hist_ms / max(args.topk_steps, 1),
                     "tie_replay_ms_per_step": replay_ms / max(args.topk_steps, 1)},
    }
    if N == 1_000_000 and q1 - q0 == N:
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic("topk")
    if world == 1 and not args.no_cpu_baseline and not tri:
        try:
            m = min(q1 - q0, 65536)
            idx, dst = t.all_pairs(k, q0, q0 + m)
            out["cpu_baseline"] = topk_cpu_baseline(Xe, k, args.cpu_seconds, idx, dst, q0)
        except AssertionError:
            raise
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    return out


def bench_sparse(args, world, rank, local, fence, data=None, steps=None, warmup=None):
    """The sparse similarity refresh (logics/item_to_item.go "users" kind over a sparse Dot collection): every item's
    top-100 neighbours by the inner product of the sqrt(idf)-weighted user sets.  A step = one all-pairs pass over this
    rank's query rows; results stay in HBM."""
    k = 100
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    if data is not None:
        desc = "S-big shard 125000 users x 200000 items x 12.5M feedbacks"
    elif args.i2i_shape == "c3":
        data, desc = synth.s_big_shard(rank=0, world=8), "S-big shard 125000 users x 200000 items x 12.5M feedbacks"
    elif args.i2i_shape == "ml1m":
        data, desc = synth.s_ml1m(), "S-ml1m 6040 users x 3706 items"
    else:
        data, desc = synth.s_ml100k(), "S-ml100k 943 users x 1682 items"
    ptr, idx, val = synth.idf_vectors(data.iptr, data.iidx, data.U)
    N = ptr.size - 1
    sp = capi.Sparse(ptr, idx, val, device=local)
    q0, q1 = gdist.shard_range(N, rank, world)
    eng = gdist.HipNeighborsEngine(sp, fetch=False)
    comm = gdist.TorchComm() if world > 1 else None
    # warm-up steps are whole steps (the pass the timed region repeats): scratch allocation, code objects, and the work plan the
    # handle keeps for its all-pairs pass (csrc/sparse.hip, gorse_sparse::Plan)
    for _ in range(max(warmup, 1)):
        gdist.refresh_neighbors_sharded(eng, comm, k, gather=False)
    fence()
    sp.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):  # gorse_amd.dist.refresh_neighbors_sharded is the function the gloo CPU tests exercise
        gdist.refresh_neighbors_sharded(eng, comm, k, gather=False)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=COLL_DEV)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    launches, ms = sp.get_profile()
    walked, hits = sp.last_stats()
    sym = sp.sym_stats()
    sp.set_profiling(False)
    if rank != 0:
        return None
    # The unit of work is the reference's: one multiply-add per (query entry, posting of its index) -- what the unsymmetric walk (and
    # the CPU baseline) performs.  The symmetric form of a full pass (round 6) meets fewer postings for the same result: `walked`.
    lens = np.bincount(idx, minlength=int(idx.max()) + 1 if idx.size else 1)
    postings = int(lens[idx[ptr[q0]:ptr[q1]]].sum())
    assert sym[0] or walked == postings, (walked, postings)
    # SURVEY 8f item 2 / DESIGN: 8 algorithmic bytes per multiply-add (the posting's accumulator id + value); the
    # accumulators themselves live in LDS
    avg_ms = ms / max(launches, 1)
    achieved = postings * 8.0 / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    out = {
        "metric": "sparse item x item top-%d multiply-adds/sec (postings of the reference's walk, whole job, N GPUs)" % k,
        "value": world * postings * steps / dt, "unit": "postings/s", "n_gpus": world, "steps": steps,
        "warmup": max(warmup, 1), "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "users item-to-item over %s: %d sparse vectors, %d entries, k=%d, query rows sharded x%d"
                               % (desc, N, int(ptr[-1]), k, world),
                   "queries_per_step_per_gpu": q1 - q0, "postings_per_step_per_gpu": postings, "postings_walked_per_step_per_gpu": walked,
                   "symmetric_pass": bool(sym[0]), "rows_redone": sym[1], "foreign_entries": sym[2], "nonzero_scores_read_back": hits},
        "roofline": {"bound": "hbm", "kernel": "sparse_tile_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_posting": 8,
                     "avg_launch_ms": avg_ms, "launches": launches},
    }
    if world == 1 and N == 200_000 and q1 - q0 == N:  # the C3-shard pass the PMC passes were taken on
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic("i2i")
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = sparse_cpu_baseline(ptr, idx, val, k, args.cpu_seconds, sp, q0)
        except AssertionError:
            raise
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "postings/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    return out


def sparse_cpu_baseline(ptr, idx, val, k, seconds, sp, q_begin):
    """The oracle's INVERTED-INDEX search (kind 'port': orc_sparse_search_inverted -- posting lists walked in ascending index
    order into per-thread accumulators, the same multiply-adds the GPU performs) on a sample of the query rows, one query per
    host thread at a time; every sampled row doubles as a bit-exact check of the GPU result."""
    from oracle import oracle as orc
    o = orc.Oracle()
    N = ptr.size - 1
    threads = min(os.cpu_count() or 1, 32)
    ix = o.sparse_index(ptr, idx, val)
    row = lambda r: (idx[ptr[r]:ptr[r + 1]], val[ptr[r]:ptr[r + 1]])
    lens = np.bincount(idx, minlength=int(idx.max()) + 1 if idx.size else 1)
    stride = 131
    sample = [q_begin + (stride * t) % max(N - q_begin, 1) for t in range(min(N - q_begin, 1 << 16))]
    # a probe sets the sample size: postings/s of one thread on the first 64 sampled rows
    scratch0 = ix.scratch()
    t0 = time.perf_counter()
    probe = sum(ix.search(*row(q), k, exclude=q, scratch=scratch0)[2] for q in sample[:64])
    rate = max(probe, 1) / max(time.perf_counter() - t0, 1e-4)
    budget, qs, acc = rate * threads * seconds, [], 0
    for q in sample:
        qs.append(q)
        acc += int(lens[row(q)[0]].sum())
        if acc >= budget:
            break
    res = [None] * len(qs)
    walked = [0] * threads

    def work(t):
        scratch = ix.scratch()
        for r in range(t, len(qs), threads):
            ei, es, w = ix.search(*row(qs[r]), k, exclude=qs[r], scratch=scratch)
            res[r] = (ei, es)
            walked[t] += w
    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    postings = int(sum(walked))
    gi, gs, gc = sp.all_pairs(k, q_begin, N)  # the pass that was timed (in its symmetric form when q_begin == 0), results fetched
    for q, (ei, es) in zip(qs, res):
        r = q - q_begin
        same = gc[r] == ei.size and np.array_equal(gi[r, :ei.size], ei) and \
            np.array_equal(gs[r, :ei.size].view(np.uint32), es.view(np.uint32))
        assert same, "GPU sparse top-k row %d differs from the oracle" % q
    return {"value": postings / dt, "unit": "postings/s", "cores": threads, "kind": "port",
            "sample": "%d query rows (every %dth from row %d) through the oracle's inverted-index search on %d threads, "
                      "%d postings walked in %.1f s; all of them compared bit for bit with the rows of the timed GPU pass"
                      % (len(qs), stride, q_begin, threads, postings, dt)}


def als_cpu_baseline(uptr, uidx, iptr, P, Q, w, reg, seconds):
    """The oracle's half-sweep (kind 'port', model.go:659-690) on a prefix of the user rows, T host threads each on
    its own row range (rows are independent: what parallel.Parallel does); the serial d x d Gram pass every call
    repeats is timed separately and subtracted."""
    from oracle import oracle as orc
    o = orc.Oracle()
    threads = min(os.cpu_count() or 1, 32)
    A = P.copy()
    t0 = time.perf_counter()
    o.als_half_range(A, Q, uptr, uidx, iptr, w, reg, 0, 0)
    t_gram = time.perf_counter() - t0
    t0 = time.perf_counter()
    o.als_half_range(A, Q, uptr, uidx, iptr, w, reg, 0, 64)
    per_entry = max(time.perf_counter() - t0 - t_gram, 1e-6) / max(int(uptr[64]), 1)
    rows = int(np.searchsorted(uptr, min(int(uptr[-1]), int(seconds * threads / per_entry)), side="right")) - 1
    rows = max(threads, min(rows, uptr.size - 1))
    cuts = [rows * t // threads for t in range(threads + 1)]

    def work(t):
        o.als_half_range(A, Q, uptr, uidx, iptr, w, reg, cuts[t], cuts[t + 1])
    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = max(time.perf_counter() - t0 - t_gram, 1e-6)
    n = int(uptr[rows])
    return {"value": n / (2.0 * dt), "unit": "entries/s", "cores": threads, "kind": "port",
            "sample": "user half-sweep over rows [0,%d) = %d feedback entries on %d threads, %.1f s after subtracting the "
                      "serial Gram pass (%.2f s, repeated per call); an epoch walks every entry twice (user + item "
                      "half-sweep), hence entries / (2 x time)" % (rows, n, threads, dt, t_gram)}


def bench_als(args, world, rank, local, fence, steps=None, warmup=None, factors=64, data=None):
    """BASELINE config C5: eALS, nFactors 64 (factors: another width on the same set -- the default line carries 16, the
    reference's default nFactors, model.go:583-596).  Every rank holds the dataset and both factor matrices and solves its
    row ranges (gorse_amd.dist.run_als_epoch); a step = one epoch = 2 half-sweeps + 2 all-gathers."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    sc = args.als_scale
    U, I, nnz, d = int(500_000 * sc), int(100_000 * sc), int(50_000_000 * sc), factors
    w, reg = 0.001, 0.06
    uptr, uidx, iptr, iidx = data if data is not None else synth.s_als(U, I, nnz, 45)
    n = int(uptr[-1])
    mf = capi.MF(U, I, d, uptr, uidx, iptr, iidx, device=local)
    P0, Q0 = synth.init_factors(U, I, d, 0.0, 0.1, seed=1)
    mf.set_factors(P0, Q0)
    comm, comm_label = make_comm(args, world, rank, local)
    eng = gdist.HipAlsEngine(mf, rank, world, staging=not isinstance(comm, gdist.LibComm))
    for _ in range(max(warmup, 1)):
        gdist.run_als_epoch(eng, comm, w, reg)
    mf.synchronize()
    fence()
    mf.set_profiling(True)
    mf.reset_profile()
    t0 = time.perf_counter()
    for _ in range(steps):
        gdist.run_als_epoch(eng, comm, w, reg)
    mf.synchronize()
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=COLL_DEV)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    ns, sweep_ms = mf.get_profile(capi.PROF_ALS_SWEEP)
    ng, gram_ms = mf.get_profile(capi.PROF_ALS_GRAM)
    mf.set_profiling(False)
    P, Q = mf.get_factors()
    if rank != 0:
        return None
    (u0, u1), (i0, i1) = eng.range
    own = int(uptr[u1] - uptr[u0]) + int(iptr[i1] - iptr[i0])  # gathered rows of this rank's two half-sweeps
    algo = own * d * 4.0 + 2.0 * ((u1 - u0) + (i1 - i0)) * d * 4  # gathers + factor rows read/written
    per_epoch_ms = (sweep_ms + gram_ms) / max(steps, 1)
    hbm_achieved = algo / (per_epoch_ms * 1e-3) / 1e9 if per_epoch_ms > 0 else 0.0
    # Primary roofline = SURVEY.md 8(d): the sweep is classed HBM-bound at 2*nnz*d*4 + 2*(U+I)*d*4 algorithmic bytes per epoch.
    # Secondary (mfma_f32_*): the Gram form this library runs does ~d times the reference recurrence's multiply-adds on the fp32
    # MFMA; it is priced on the STRICT upper triangle of q q^T (d (d + 1) / 2 multiply-adds per gathered row), not on the three
    # full 32 x 32 blocks the kernel actually issues at nFactors 64 -- implementation work is not algorithmic work.
    macs_per_row = d * (d + 1) // 2
    flops = 2.0 * own * macs_per_row
    mfma_achieved = flops / (per_epoch_ms * 1e-3) / 1e12 if per_epoch_ms > 0 else 0.0
    out = {
        "metric": "ALS feedback entries/sec (nnz per epoch x epochs / time, whole job, N GPUs)",
        "value": n * steps / dt, "unit": "entries/s", "n_gpus": world, "steps": steps,
        "warmup": max(warmup, 1), "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "S-als %dx%dx%d (C5%s), nFactors=%d, weight=%g reg=%g" % (U, I, n, "" if sc == 1.0 else " x%g" % sc, d, w, reg),
                   "parallelism": "rows sharded x%d, factors replicated, 2 all-gathers((U+I)*d fp32)/epoch over %s" % (world, comm_label)
                   if world > 1 else "single GPU", "factors_finite": bool(np.isfinite(P).all() and np.isfinite(Q).all())},
        "roofline": {"bound": "hbm", "kernel": ("als_wide_kernel + als_gram_partial_kernel" if d > 64 else "als_row_kernel + als_chunk_kernel") + " (+ S Gram)",
                     "achieved": hbm_achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_epoch": algo, "avg_launch_ms": per_epoch_ms, "launches": ns,
                     "mfma_f32_tflops": mfma_achieved, "mfma_f32_frac": mfma_achieved / MFMA_F32_PEAK_TFLOPS,
                     "mfma_macs_per_gathered_row": macs_per_row,
                     "flop_note": "secondary figure: 2 flop x gathered rows of both half-sweeps x d (d + 1) / 2 multiply-adds (strict "
                                  "upper triangle of the Gram update) against the %.1f TFLOP/s fp32 MFMA peak; at nFactors 32 / 64 each "
                                  "multiply-add is formed as six exact bf16-MFMA partial products of three-way split floats, summed in "
                                  "fp32 (csrc/als.hip gram_accumulate_b3) -- the figure stays in fp32 multiply-adds" % MFMA_F32_PEAK_TFLOPS,
                     "gram_products": "bf16x3 split, fp32 accumulate" if d in (32, 64) or d > 64 else "fp32 MFMA",
                     "sweeps_ms_per_epoch": sweep_ms / max(steps, 1), "gram_ms_per_epoch": gram_ms / max(steps, 1)},
    }
    if world == 1 and sc == 1.0 and d == 64:
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic("als")
    if world == 1 and not args.no_cpu_baseline and d == 64:
        try:
            out["cpu_baseline"] = als_cpu_baseline(uptr, uidx, iptr, P0, Q0, w, reg, args.cpu_seconds)
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "entries/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    return out


def make_comm(args, world, rank, local):
    """The exchange path of a multi-rank run: the library's own RCCL communicator (csrc/comm.hip; the unique id travels from
    rank 0 through a torch.distributed broadcast), or torch.distributed's with --comm torch / when the library's cannot be
    set up.  Returns (comm, label)."""
    if world == 1:
        return None, "single GPU"
    label = "torch.distributed " + BACKEND
    if args.comm == "lib" and BACKEND == "nccl":
        lib, why = None, ""
        try:
            def share(uid):
                t = torch.tensor(list(uid), dtype=torch.uint8, device=COLL_DEV)
                dist.broadcast(t, src=0)
                return bytes(t.cpu().tolist())
            def agree(ok):
                t = torch.tensor([1.0 if ok else 0.0], device=COLL_DEV)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return float(t.item()) > 0
            lib = gdist.LibComm(rank, world, local, share, agree=agree)
        except Exception as e:  # reported in the line; the run goes on over torch.distributed
            why = repr(e)
        ok = torch.tensor([1.0 if lib is not None else 0.0], device=COLL_DEV)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # every rank takes the same path
        if float(ok.item()) > 0:
            return lib, "gorse_comm (RCCL inside libgorse_hip, collectives on the handle's stream)"
        label = "torch.distributed nccl (gorse_comm unavailable on some rank%s)" % (": " + why if why else "")
    return gdist.TorchComm(), label


def bench_bpr(args, workload, world, rank, local, comm, comm_label, steps, warmup, data=None, with_cpu=True, factors=0):
    """One BPR leg: a step = one epoch = CountFeedback() samples over this rank's resident shard + the item-factor exchange."""
    if data is None:
        data = make_data(workload, rank, args)[0]
    d, desc = make_data_desc(workload, factors or args.factors)
    lr, reg = 0.05, 0.01
    n_samples = data.n_train
    if workload in ("big", "c3full"):  # what ran, to the feedback
        desc += ", %d feedbacks = %.1f per user" % (n_samples, n_samples / data.U)
    t_create = time.perf_counter()
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx, device=local)
    t_create = time.perf_counter() - t_create
    init = synth.init_factors_big if data.U * d >= 1 << 26 else synth.init_factors
    P0, Q0 = init(data.U, data.I, d, 0.0, 0.001, seed=1)  # same Q on every rank
    if world > 1:
        P0 = synth.init_factors(data.U, 1, d, 0.0, 0.001, seed=100 + rank)[0]
    mf.set_factors(P0, Q0)
    del P0, Q0
    engine = gdist.HipEngine(mf, args.mode)
    if world > 1:
        if isinstance(comm, gdist.LibComm):
            mf.item_sync_mark()
        else:
            engine.enable_exchange()

    def step(epoch):  # gorse_amd.dist.run_epoch is the function the gloo CPU tests exercise
        gdist.run_epoch(engine, comm, n_samples, lr, reg, 2024, epoch, rank * (1 << 40))

    def fence():
        mf.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for w in range(warmup):
        step(w + 1)
    fence()
    mf.set_profiling(True)
    mf.reset_profile()
    t0 = time.perf_counter()
    for s in range(steps):
        step(warmup + s + 1)
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=COLL_DEV)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    launches, upd_ms = mf.get_profile(capi.PROF_BPR_UPDATE)
    s_launches, smp_ms = mf.get_profile(capi.PROF_BPR_SAMPLE)
    o_launches, sort_ms = mf.get_profile(capi.PROF_BPR_SORT)
    c_launches, comm_ms = mf.get_profile(capi.PROF_COMM)
    user_runs = args.mode in (capi.BPR_HOGWILD_ATOMIC, capi.BPR_HOGWILD_STORES) and mf.bpr_user_runs()
    mf.set_profiling(False)
    # the same epochs through the SYNCHRONOUS entry point (gorse_bpr_epoch: what a Fit loop that evaluates after every epoch
    # calls): each call drains both streams, so nothing of epoch e + 1 is prepared under epoch e
    sync_ms = None
    if world == 1 and args.sync_steps > 0 and args.mode != capi.BPR_SEQUENTIAL:
        t0 = time.perf_counter()
        for s in range(args.sync_steps):
            mf.bpr_epoch(n_samples, lr, reg, 2024, warmup + steps + s + 1, mode=args.mode)
        sync_ms = (time.perf_counter() - t0) / args.sync_steps * 1e3
    P, Q = mf.get_factors()
    finite = bool(np.isfinite(P).all() and np.isfinite(Q).all())
    del P, Q
    mf.close()
    if rank != 0:
        return None
    bytes_per_sample = 6 * d * 4 + 12  # three rows read + three written + indices
    samples_per_launch = steps * n_samples / max(launches, 1)
    avg_ms = upd_ms / max(launches, 1)
    achieved = samples_per_launch * bytes_per_sample / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    ws_mb = (data.U + data.I) * d * 4 / 1e6
    out = {
        "metric": "BPR positive-samples/sec (whole job, N GPUs)",
        "value": world * n_samples * steps / dt,
        "unit": "samples/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc, "samples_per_step_per_gpu": n_samples, "lr": lr, "reg": reg,
                   "schedule": {0: "hogwild-atomic", 1: "sequential", 2: "hogwild-racy", 3: "hogwild-stores (GORSE_BPR_HOGWILD_STORES: what Fit runs)"}[args.mode]
                   + (", user runs (sample ids counting-sorted by user, p_u register-resident, cold negatives by store)" if user_runs else
                      (", one group per sample" if args.mode in (capi.BPR_HOGWILD_ATOMIC, capi.BPR_HOGWILD_STORES) else "")),
                   "entry_point": "gorse_bpr_epoch_enqueue x steps, one synchronisation at the end (a Fit between two evaluations)",
                   "sync_entry_point_ms_per_step": sync_ms,
                   "parallelism": "users sharded x%d, item factors replicated, 1 all-reduce(%.1f MB = I*d fp32)/epoch over %s"
                   % (world, data.I * d * 4 / 1e6, comm_label) if world > 1 else "single GPU", "factors_finite": finite,
                   "handle_create_seconds": t_create},
        "roofline": {"bound": "hbm", "kernel": "bpr_update_user_kernel" if user_runs else "bpr_update_kernel",
                     "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_sample": bytes_per_sample, "avg_launch_ms": avg_ms,
                     "launches": launches, "samples_per_launch": samples_per_launch,
                     "sampler_avg_ms": smp_ms / max(s_launches, 1),
                     "user_sort_avg_ms": sort_ms / max(o_launches, 1) if o_launches else 0.0,
                     "note": "working set P + Q = %.1f MB (%s the 256 MiB Infinity Cache)" % (ws_mb, "inside" if ws_mb < 268 else "outside")},
    }
    if user_runs and ws_mb < 32.0 and avg_ms > 0:
        # The working set (P + Q) never leaves the L2s (4 MB x 8 XCDs) / the Infinity Cache: "fraction of HBM peak" describes nothing
        # physical here.  What the launch IS bound by: every item update is d fp32 atomics performed memory-side (at the L2 / fabric),
        # two item rows per sample, against the rate the unit gives under THIS access mix -- a table of this many rows that is also
        # gathered two samples ahead: scripts/probe_atomics3.hip, profiles/r04_s_probe_atomics3.txt: 243 G dword/s (318 G when nothing
        # reads the rows; DESIGN.md section 4 "Why C2 stays").  The nominal HBM figure stays next to it (`frac`).
        atomic_dw = samples_per_launch * 2 * d
        a_rate = atomic_dw / (avg_ms * 1e-3) / 1e9
        out["roofline"]["bound_measured"] = "l2_atomic"
        out["roofline"]["l2_atomic"] = {"achieved": a_rate, "peak": 243.0, "unit": "G atomic dwords/s", "frac": a_rate / 243.0,
                                        "atomic_dwords_per_sample": 2 * d,
                                        "note": "peak = fp32 atomic adds/s the memory side sustains on a 3,704-row table whose rows are also "
                                                "gathered two iterations ahead (scripts/probe_atomics3.hip, profiles/r04_s_probe_atomics3.txt); "
                                                "318 G/s when nothing reads the rows"}
    if world > 1:
        out["exchange"] = {"allreduce_avg_ms": comm_ms / max(c_launches, 1) if c_launches else None, "launches": c_launches,
                           "bytes": data.I * d * 4, "path": comm_label}
    if args.mode in (capi.BPR_HOGWILD_ATOMIC, capi.BPR_HOGWILD_STORES) and workload in ("ml1m", "c3", "c3full", "big") and not (factors or args.factors):
        key = {"ml1m": "ml1m_users" if user_runs else "ml1m", "c3": "c3_users", "c3full": "c3full_users", "big": "big_users"}[workload]
        tr, src = measured_traffic(key)
        out["roofline"]["traffic_session_round"] = TRAFFIC_ROUND if tr is not None else None
        # the recorded figure is the mean over the FULL-size launches (one chunk: 32 samples per user, clamped to [4M, 128M], or the
        # whole epoch); `traffic` is scaled to the mean launch samples_per_launch describes, the full-size figure rides along
        full_launch = min(n_samples, min(max(32 * data.U, 4 << 20), 128 << 20))
        out["roofline"]["traffic_full_launch"], out["roofline"]["full_launch_samples"] = tr, full_launch
        out["roofline"]["traffic"] = tr * samples_per_launch / full_launch if tr is not None else None
        out["roofline"]["traffic_source"] = src
    if world == 1 and with_cpu and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(data, d, lr, reg, args.cpu_seconds if with_cpu is True else float(with_cpu))
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": 0, "kind": "port",
                                   "sample": "failed: %r" % (e,)}
    return out


def bench_fit(args, local):
    """The reference's own training shape, end to end (model/cf/model_test.go:35-48: ml-1m, nFactors 8, 30 epochs, lr .05, reg .01,
    init N(0, .001); and the default width 16, model.go:394) through the C++ twin of BPR.Fit (gorse_amd/host/gorse_cf.cpp): Init,
    the epoch loop with FitConfig's default Verbose 10 (evaluations before epoch 1 and after 10, 20, 30: Rank + TopKFilter on the
    device over the NCF layout's 100 candidates per user), Jobs = host cores -> the Hogwild schedule.  Reports the wall time of
    Fit, the mean epoch and evaluation times of its own log (fit_time / eval_time, model.go:496-503) and NDCG@10 next to the
    sequential oracle's on the same data (different init draws and sampler stream: the comparison is +-0.01, like the
    reference's own test around its anchor)."""
    import re
    from gorse_amd import cf
    data = synth.s_ml1m()
    train, test = cf.datasets_from_synth(data)
    epochs, lr, reg = 30, 0.05, 0.01
    out = {"metric": "BPR.Fit wall seconds (S-ml1m, 30 epochs, Verbose 10)", "unit": "s", "higher_is_better": False,
           "data": "synthetic", "config": {"workload": "S-ml1m 6040x3706x%d, model_test.go:35-48 hyper-parameters" % data.n_train}}
    for d in (8, 16):
        m = cf.NewBPR({"NFactors": d, "Reg": reg, "Lr": lr, "NEpochs": epochs, "InitMean": 0, "InitStdDev": 0.001})
        cfg = cf.NewFitConfig().SetJobs(max(2, os.cpu_count() or 2))
        m.Fit(train, test, cfg)  # first Fit of a width: code objects, buffers
        m = cf.NewBPR({"NFactors": d, "Reg": reg, "Lr": lr, "NEpochs": epochs, "InitMean": 0, "InitStdDev": 0.001, "RandomState": 1})
        t0 = time.perf_counter()
        score = m.Fit(train, test, cfg)
        wall = time.perf_counter() - t0
        fit_ms = [float(x) for x in re.findall(r"fit_time=([0-9.]+)ms", m.log)]
        eval_ms = [float(x) for x in re.findall(r"eval_time=([0-9.]+)ms", m.log)]
        rec = {"value": wall, "epochs_done": m.epochs_done, "fit_ms_per_epoch": float(np.mean(fit_ms)) if fit_ms else None,
               "samples_per_s": data.n_train / (float(np.mean(fit_ms)) * 1e-3) if fit_ms else None,
               "eval_ms": float(np.mean(eval_ms)) if eval_ms else None, "evaluations": len(eval_ms), "ndcg": score.NDCG}
        if fit_ms:  # SURVEY 8(d) bytes per sample over the epoch's DEVICE time (fit_time of the twin's log = gorse_mf_epoch_times)
            ep_s = float(np.mean(fit_ms)) * 1e-3
            gbs = data.n_train * (6 * d * 4 + 12) / ep_s / 1e9
            rec["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                               "l2_atomic_frac": data.n_train * 2 * d / ep_s / 1e9 / 243.0}  # of the 243 G atomic dwords/s (see the headline)
        if not args.no_cpu_baseline:
            from oracle import oracle as orc
            o = orc.Oracle()
            P, Q = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 3)
            srt = orc.sort_rows(data.uptr, data.uidx)
            t0 = time.perf_counter()
            for ep in range(1, epochs + 1):
                o.bpr_epoch_sampled(P, Q, data.uptr, data.uidx, srt, 77, ep, 0, data.n_train, lr, reg)
            rec["oracle_ndcg"] = float(o.evaluate(P, Q, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0])
            rec["oracle_seconds"] = time.perf_counter() - t0
        out["d%d" % d] = rec
    out["value"] = out["d8"]["value"]
    # ALS.Fit at the reference's own shape (model/cf/model_test.go:93-104: ml-1m, nFactors 8, Reg 0.015, Alpha 0.05, 30 epochs; and
    # the default width 16, model.go:583-596) through the C++ twin of ALS.Fit, same FitConfig; ms per epoch from its own log, NDCG@10
    # next to the oracle's (the reference's float32 recurrence, 30 epochs from the same kind of init)
    reg_a, alpha = 0.015, 0.05
    for d in (8, 16):
        params = {"NFactors": d, "Reg": reg_a, "Alpha": alpha, "NEpochs": epochs}
        m = cf.NewALS(dict(params))
        cfg = cf.NewFitConfig().SetJobs(max(2, os.cpu_count() or 2))
        m.Fit(train, test, cfg)  # first Fit of a width: code objects, buffers, the row plan
        m = cf.NewALS(dict(params, RandomState=1))
        t0 = time.perf_counter()
        score = m.Fit(train, test, cfg)
        wall = time.perf_counter() - t0
        fit_ms = [float(x) for x in re.findall(r"fit_time=([0-9.]+)ms", m.log)]
        eval_ms = [float(x) for x in re.findall(r"eval_time=([0-9.]+)ms", m.log)]
        nnz = int(data.uptr[-1])
        rec = {"value": wall, "epochs_done": m.epochs_done, "fit_ms_per_epoch": float(np.mean(fit_ms)) if fit_ms else None,
               "entries_per_s": nnz / (float(np.mean(fit_ms)) * 1e-3) if fit_ms else None,
               "eval_ms": float(np.mean(eval_ms)) if eval_ms else None, "evaluations": len(eval_ms), "ndcg": score.NDCG}
        if fit_ms:  # 2 nnz d 4 + 2 (U + I) d 4 bytes per epoch
            algo = 2.0 * nnz * d * 4 + 2.0 * (data.U + data.I) * d * 4
            rec["hbm_frac"] = algo / (float(np.mean(fit_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS
            rec["roofline"] = {"bound": "hbm", "achieved": rec["hbm_frac"] * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": rec["hbm_frac"], "traffic": None}  # (avg_launch_ms = fit_ms_per_epoch)
        if not args.no_cpu_baseline:
            from oracle import oracle as orc
            o = orc.Oracle()
            P, Q = synth.init_factors(data.U, data.I, d, 0.0, 0.1, 3)
            t0 = time.perf_counter()
            for ep in range(epochs):
                P, Q = o.als_epoch(P, Q, data.uptr, data.uidx, data.iptr, data.iidx, alpha, reg_a)
            rec["oracle_ndcg"] = float(o.evaluate(P, Q, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0])
            rec["oracle_seconds"] = time.perf_counter() - t0
        out["als_d%d" % d] = rec
    return out


def bench_mm(args, local):
    """floats.MM (common/floats/floats.go:241: C += A B as an l-ascending fmaf chain per element) on the fp32 MFMA, where the
    reference's CTR models and its tests call it.  4096^3, NN; the kernel time comes from hipEvents around the launch inside
    gorse_hip_sgemm (the entry point takes host buffers: the copies are not part of the figure; PCIe-inclusive: `host_to_host_ms`)."""
    n = 4096
    rng = np.random.default_rng(3)
    a = rng.standard_normal((n, n)).astype(np.float32)
    b = rng.standard_normal((n, n)).astype(np.float32)
    c0 = np.zeros((n, n), np.float32)
    L = capi.lib()
    capi.sgemm(0, 0, n, n, n, a.ravel(), n, b.ravel(), n, c0.ravel(), n, device=local)  # code objects
    ms, wall = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        capi.sgemm(0, 0, n, n, n, a.ravel(), n, b.ravel(), n, c0.ravel(), n, device=local)
        wall.append((time.perf_counter() - t0) * 1e3)
        ms.append(L.gorse_hip_test_sgemm_last_ms())
    t = float(np.median(ms))
    tf = 2.0 * n * n * n / (t * 1e-3) / 1e12
    return {"metric": "floats.MM TFLOP/s (fp32, 4096^3 NN, bit-equal to the reference's fmaf chain)", "value": tf, "unit": "TFLOP/s",
            "config": {"workload": "floats.MM 4096 x 4096 x 4096 fp32, NN (common/floats/mm.go:19-49)"},
            "ms_per_step": t, "steps": 3, "higher_is_better": True, "dtype": "f32", "data": "synthetic", "host_to_host_ms": float(np.median(wall)),
            "roofline": {"bound": "mfma_f32", "kernel": "sgemm_mfma_kernel", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": measured_traffic("mm")[0], "traffic_source": measured_traffic("mm")[1],
                         "avg_launch_ms": t, "launches": 3}}


def leg(fn, what):
    """a secondary leg of the default line: its failure costs the line that object only"""
    try:
        return fn()
    except AssertionError:
        raise  # a parity check inside a leg failed: that must not pass silently
    except Exception as e:
        return {"metric": what, "value": None, "error": repr(e)}


def emit(out, tag="default"):
    """ONE JSON line: the headline keys first, then `topk` (the second half of BASELINE.json's metric), then the other
    configurations; prose goes to the notes file (see compact)."""
    head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "topk"]
    ordered = {k: out[k] for k in head if k in out}
    ordered.update({k: v for k, v in out.items() if k not in ordered})
    notes = {}
    line = compact(ordered, notes)
    # the driver's tail is 8 KB: if the line is still long, the secondary objects' `config` blocks move to the notes as well
    # the secondary objects first lose what the headline already says about the whole line (one GPU, synthetic data, ...) and
    # the finer points of their rooflines: moved to the notes, numbers included
    boiler = ("n_gpus", "warmup", "higher_is_better", "scaling", "vs_baseline", "data")
    fine = ("algorithmic_bytes_per_sample", "sampler_avg_ms", "user_sort_avg_ms", "algorithmic_flop_per_pair", "mfma_macs_per_gathered_row",
            "algorithmic_bytes_per_epoch", "sweeps_ms_per_epoch", "gram_ms_per_epoch", "full_launch_samples", "traffic_full_launch",
            "traffic_session_round")
    if isinstance(line.get("roofline"), dict):  # the headline keeps its figures but for the bookkeeping of the traffic record
        notes["roofline.moved"] = {f: line["roofline"].pop(f) for f in ("full_launch_samples", "traffic_full_launch", "traffic_session_round")
                                   if f in line["roofline"]}
    if isinstance(line.get("fit"), dict):  # per Fit record: what follows from the figures that stay, or describes the check beside them
        for k, v in line["fit"].items():
            if isinstance(v, dict):
                moved = {f: v.pop(f) for f in ("epochs_done", "evaluations", "oracle_seconds", "samples_per_s", "entries_per_s", "hbm_frac") if f in v}
                if moved:
                    notes["fit.%s.moved" % k] = moved
    # what a secondary object's roofline keeps in the line itself (the contract's keys + the stage times of the top-k pass); the rest
    # -- launch counts, samples per launch, clock, secondary rates -- rides in the notes, numbers included
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "mfma_f32_frac", "rescore_avg_ms",
            "tie_history_sweep_ms_per_step", "tie_replay_ms_per_step", "bound_measured", "l2_atomic")
    for k, v in line.items():
        if k in ("config", "roofline", "cpu_baseline") or not isinstance(v, dict) or "metric" not in v:
            continue
        moved = {b: v.pop(b) for b in boiler if b in v}
        if isinstance(v.get("metric"), str) and " (" in v["metric"]:  # the parenthetical of a secondary metric's name: to the notes
            moved["metric"] = v["metric"]
            v["metric"] = v["metric"].split(" (")[0]
        if isinstance(v.get("roofline"), dict):
            moved.update({"roofline." + f: v["roofline"].pop(f) for f in list(v["roofline"]) if f in fine or f not in keep})
        if moved:
            notes[k + ".moved"] = moved
    # (what is left of a moved block is its `workload` string: every object says in the line itself what it ran)
    for k in ("mm", "ml100k_d8", "ml100k", "als_d128", "als_d16", "i2i", "als", "c3", "big", "fit"):
        if len(json.dumps(line, separators=(",", ":"))) < 7000:
            break
        if isinstance(line.get(k), dict) and isinstance(line[k].get("config"), dict) and len(line[k]["config"]) > 1:
            notes[k + ".config"] = line[k]["config"]
            line[k]["config"] = {"workload": line[k]["config"].get("workload")}
    # last resort for the 8 KB tail: the rooflines of the `fit` records (derived from their fit_ms_per_epoch) move to the notes too
    if len(json.dumps(line, separators=(",", ":"))) >= 7900 and isinstance(line.get("fit"), dict):
        for k, v in line["fit"].items():
            if isinstance(v, dict) and "roofline" in v:
                notes["fit.%s.roofline" % k] = v.pop("roofline")
    # the notes of this run lie next to the line's log (gpurun_out/ is scratch); the copy of the round's last run is committed as
    # profiles/r06_bench_notes_<tag>.json
    line["notes"] = {"this_run": write_notes(notes, tag), "committed_copy": "profiles/r06_bench_notes_%s.json" % tag}
    print(json.dumps(line, separators=(",", ":")), flush=True)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if BACKEND != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if BACKEND == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(BACKEND)
    if args.gpus != world:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU path exists)"
    torch.cuda.set_device(local)
    # BASELINE.json quotes the single-GPU number on C2 (S-ml1m) and the 8-GPU number on C3: that is what N = 1 / N > 1 run
    workload = args.workload or ("ml1m" if world == 1 else "c3")
    if args.bpr_chunk:
        capi.lib().gorse_hip_test_set_bpr_chunk(args.bpr_chunk)

    def fence0():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if workload in ("topk", "als", "i2i"):
        if workload == "als":
            out = bench_als(args, world, rank, local, fence0)
        elif workload == "i2i":
            out = bench_sparse(args, world, rank, local, fence0)
        else:
            out = bench_topk(args, world, rank, local, fence0)
            if rank == 0:
                out["warmup"] = 1
                out["vs_baseline"] = None
        if rank == 0:
            emit(out, workload)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    comm, comm_label = make_comm(args, world, rank, local)
    steps, warmup = args.steps, args.warmup
    if workload in ("c3full", "big") and args.steps == 20 and args.warmup == 3:  # the defaults are sized for a 0.7 ms epoch
        steps, warmup = (5, 2) if workload == "c3full" else (3, 1)
    out = bench_bpr(args, workload, world, rank, local, comm, comm_label, steps, warmup)
    full_line = args.workload is None and world == 1  # the default single-GPU run carries the other configurations too
    # BASELINE.json's metric has two halves; the second one rides on the C2 line of one GPU and on the C3 line of N > 1
    # (query rows sharded over the ranks, no collective)
    if ((workload == "ml1m" and world == 1) or (workload == "c3" and world > 1 and args.workload is None)) and not args.no_topk:
        # N > 1: the top-k leg's exchanges (the triangle shard's all-gather and all-to-all over RCCL) have run on emulated ranks and over
        # gloo only -- no multi-GPU node was ever available to the builder.  A collective that never completes must not cost the line
        # its BPR half: a watchdog on every rank emits the line WITHOUT the leg (rank 0) and leaves, should the leg not return in time.
        dog = None
        if world > 1:
            def bark():
                if rank == 0:
                    out["topk"] = {"metric": "item x item cosine top-100 pairs/sec", "value": None,
                                   "error": "the sharded top-k leg did not return within %d s (watchdog); --topk-shard rows has no exchange" % args.topk_watchdog}
                    emit(out, args.workload or "default_n%d" % world)
                os._exit(0)
            dog = threading.Timer(args.topk_watchdog, bark)
            dog.daemon = True
            dog.start()
        topk = leg(lambda: bench_topk(args, world, rank, local, fence0), "item x item cosine top-100 pairs/sec")
        if dog is not None:
            dog.cancel()
        if rank == 0:
            out["topk"] = topk
    if full_line and not args.no_extra:
        full = synth.s_big_full()
        out["c3"] = leg(lambda: bench_bpr(args, "c3full", 1, 0, local, None, "single GPU", 5, 2, data=full, with_cpu=5.0),
                        "BPR positive-samples/sec, C3 whole on one GPU")
        big = shard_of(full, 0, 8)
        del full
        out["i2i"] = leg(lambda: bench_sparse(args, 1, 0, local, fence0, data=big, steps=3, warmup=1), "sparse item x item top-100")
        del big
        s_als = synth.s_als(int(500_000 * args.als_scale), int(100_000 * args.als_scale), int(50_000_000 * args.als_scale), 45)
        out["als"] = leg(lambda: bench_als(args, 1, 0, local, fence0, steps=3, warmup=1, data=s_als), "ALS feedback entries/sec")
        out["als_d16"] = leg(lambda: bench_als(args, 1, 0, local, fence0, steps=3, warmup=1, factors=16, data=s_als),
                             "ALS feedback entries/sec, C5's set at the reference's default nFactors 16")
        # 65 <= nFactors <= 128 take another set of kernels (als_wide_kernel: M in LDS, a workgroup per row): C5's set at the widest
        out["als_d128"] = leg(lambda: bench_als(args, 1, 0, local, fence0, steps=2, warmup=1, factors=128, data=s_als),
                              "ALS feedback entries/sec, C5's set at nFactors 128")
        del s_als
        # the reference's own hyper-parameters (model/cf/model_test.go:35-45: nFactors 16; 8 is the smallest it searches)
        out["ml100k"] = leg(lambda: bench_bpr(args, "ml100k", 1, 0, local, None, "single GPU", 20, 3, with_cpu=False),
                            "BPR positive-samples/sec, S-ml100k nFactors 16")
        out["ml100k_d8"] = leg(lambda: bench_bpr(args, "ml100k", 1, 0, local, None, "single GPU", 20, 3, with_cpu=False, factors=8),
                               "BPR positive-samples/sec, S-ml100k nFactors 8")
        out["mm"] = leg(lambda: bench_mm(args, local), "floats.MM TFLOP/s")
        out["fit"] = leg(lambda: bench_fit(args, local), "BPR.Fit wall seconds (S-ml1m, nFactors 8 / 16, 30 epochs)")
        # north_star's 10M x 1M x 128 set: P = 5.1 GB, far outside every cache.  250M draws (220M distinct feedbacks, 22 per user)
        # instead of --workload big's 1.25e9 so that the set generates in seconds inside the driver's run; same users, same items
        out["big"] = leg(lambda: bench_bpr(args, "big", 1, 0, local, None, "single GPU", 3, 1, data=synth.s_huge(N=BIG_DRAWS_DEFAULT_LINE),
                                           with_cpu=5.0), "BPR positive-samples/sec, 10M users x 1M items, nFactors 128")
    if rank == 0:
        emit(out, args.workload or ("default" if world == 1 else "default_n%d" % world))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
