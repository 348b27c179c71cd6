#!/usr/bin/env python3
"""GPU probe (probe build: make -C gorse_amd/csrc probe-lib): the software pipeline of the atomics-only user-run BPR kernel --
item rows gathered G samples ahead, their indices IA ahead (csrc/bpr.hip) -- per shape: update-kernel ms per epoch and NDCG@10.
The counter the kernel's waits go by returns in order and counts the atomics too, so a load waits behind the atomics issued
before it; with few groups per SIMD (C2: 6040 groups on 1024 SIMDs) the distance min(G, IA - G) sets the time of an iteration.

usage: gpu_probe_bpr_depth.py [c2] [c2d16] [c2d8] [c3s]      Output -> profiles/rNN_*_probe_bpr_depth.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_probe = os.path.join(ROOT, "gorse_amd", "lib", "libgorse_hip_probe.so")
if os.path.exists(_probe) and os.path.getmtime(_probe) >= os.path.getmtime(os.path.join(ROOT, "gorse_amd", "lib", "libgorse_hip.so")):
    os.environ.setdefault("GORSE_HIP_LIB", _probe)
import numpy as np

from gorse_amd import capi, synth
from oracle import oracle as orc

L = capi.lib()
o = orc.Oracle()
args = sys.argv[1:] or ["c2"]
DEPTHS = [(0, "shipped"), (3, "G=3 IA=6"), (6, "G=6 IA=9"), (10, "ring 3/1"), (11, "ring 4/2"), (12, "ring 6/3"), (13, "ring 8/4"), (14, "ring 6/2")]
print("probe build:", bool(L.gorse_hip_test_probe_build()), flush=True)


def run_case(name, data, d, epochs, store):
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 1)
    L.gorse_hip_test_set_bpr_store_mode(store)
    L.gorse_hip_test_set_variant(128)
    for which, label in DEPTHS:
        L.gorse_hip_test_set_bpr_user_depth(which)
        mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
        mf.set_factors(P0, Q0)
        mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 9)
        mf.synchronize()
        mf.set_factors(P0, Q0)
        mf.set_profiling(True)
        mf.reset_profile()
        t0 = time.perf_counter()
        for e in range(epochs):
            mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 77, 1 + e)
        mf.synchronize()
        wall = time.perf_counter() - t0
        _, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
        mf.set_profiling(False)
        gp, gq = mf.get_factors()
        ndcg = float("nan")
        if np.isfinite(gp).all() and np.isfinite(gq).all() and data.test_idx.size:
            ndcg = o.evaluate(gp, gq, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0]
        print("%-6s d=%3d %-10s update %8.4f ms/epoch (%.3e samples/s)  wall %8.4f ms/epoch  NDCG %.4f" % (
            name, d, label, ms / epochs, epochs * data.n_train / (ms * 1e-3), wall / epochs * 1e3, ndcg), flush=True)
        mf.close()
    L.gorse_hip_test_set_bpr_user_depth(0)
    L.gorse_hip_test_set_bpr_store_mode(-1)
    L.gorse_hip_test_set_variant(0)


if "c2" in args:
    run_case("c2", synth.s_ml1m(), 64, 8, 0)
if "c2d16" in args:
    run_case("c2d16", synth.s_ml1m(), 16, 8, 0)
if "c2d8" in args:
    run_case("c2d8", synth.s_ml1m(), 8, 8, 0)
if "c2d128" in args:
    run_case("c2d128", synth.s_ml1m(), 128, 8, 0)
if "c3s" in args:
    run_case("c3s", synth.hold_out(synth.s_big_shard(rank=0, world=8), 8192, 99, 5), 128, 1, 0)
