#!/usr/bin/env python3
"""GPU probe (probe build): what bounds the C2 epoch (S-ml1m, 6040 user runs, d = 64) -- the ring kernel (csrc/bpr.hip) at 256 / 128 /
64 threads per workgroup, without its item updates (the cost of everything else), and on a data set of the same shape whose items
are equally popular (zipf_s = 0: no hot rows, no replicas), with the item count varied (the footprint of the atomics).

usage: gpu_probe_bpr_c2_limits.py      Output -> profiles/rNN_*_probe_bpr_c2_limits.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_probe = os.path.join(ROOT, "gorse_amd", "lib", "libgorse_hip_probe.so")
if os.path.exists(_probe) and os.path.getmtime(_probe) >= os.path.getmtime(os.path.join(ROOT, "gorse_amd", "lib", "libgorse_hip.so")):
    os.environ.setdefault("GORSE_HIP_LIB", _probe)
import numpy as np

from gorse_amd import capi, synth

L = capi.lib()
print("probe build:", bool(L.gorse_hip_test_probe_build()), flush=True)
NOATOM = 1 << 24


def run(label, data, d, depth, variant, epochs=8):
    P0, Q0 = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 1)
    L.gorse_hip_test_set_bpr_store_mode(0)
    L.gorse_hip_test_set_variant(128 | variant)
    L.gorse_hip_test_set_bpr_user_depth(depth)
    mf = capi.MF(data.U, data.I, d, data.uptr, data.uidx)
    mf.set_factors(P0, Q0)
    mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 9)
    mf.synchronize()
    mf.set_factors(P0, Q0)
    mf.set_profiling(True)
    mf.reset_profile()
    for e in range(epochs):
        mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 77, 1 + e)
    mf.synchronize()
    _, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
    mf.set_profiling(False)
    gp, gq = mf.get_factors()
    print("%-58s d=%3d update %8.4f ms/epoch  %.3e samples/s  %6.1f G atomic dwords/s  finite %s" % (
        label, d, ms / epochs, epochs * data.n_train / (ms * 1e-3), epochs * data.n_train * 2 * d / (ms * 1e-3) / 1e9,
        bool(np.isfinite(gp).all() and np.isfinite(gq).all())), flush=True)
    mf.close()
    L.gorse_hip_test_set_bpr_user_depth(0)
    L.gorse_hip_test_set_variant(0)
    L.gorse_hip_test_set_bpr_store_mode(-1)


ml = synth.s_ml1m()
for d in (64, 16, 128):
    run("S-ml1m, rotating kernel (shipped)", ml, d, 0, 0)
    for gpw in (4, 2, 1):
        run("S-ml1m, ring 3/1, %d of 4 groups of a wave working" % gpw, ml, d, 10 | (256 << 8) | (gpw << 20), 0)
        run("S-ml1m, ring 3/1, %d of 4 groups, NO item updates" % gpw, ml, d, 10 | (256 << 8) | (gpw << 20), NOATOM)
    run("S-ml1m, ring 6/3, 1 of 4 groups", ml, d, 12 | (256 << 8) | (1 << 20), 0)
WARM_ELSEWHERE = 1 << 23
for d in (64, 16):
    run("S-ml1m, shipped kernel, atomics of the items without replicas onto a scratch table (TIMING ONLY)", ml, d, 0, WARM_ELSEWHERE)
for items in (3706, 200000):
    uni = synth.synth_cf(6040, items, 994169, seed=42, zipf_s=0.0, min_len=19, n_neg=99, with_test=False)
    for gpw in (4, 1):
        run("uniform items, I=%d, ring 3/1, %d of 4 groups, no replicas" % (items, gpw), uni, 64, 10 | (256 << 8) | (gpw << 20), 32)
