#!/usr/bin/env python3
"""GPU probe: the user-run BPR update with write-through stores on cold item rows (csrc/bpr.hip ST_*), against the atomics-only
form, per shape: update-kernel ms per epoch, end-to-end rate, NDCG@10 next to the sequential oracle, and the lost updates.

Lost updates are counted, not guessed: with P = 2^-10 everywhere, Q = 0, reg = 0 and lr = 2^-20 every sample moves its positive
row by +2^-31 and its negative row by -2^-31 in every coordinate (grad = 1/2 to 5 digits), so after one epoch
Q[r][0] / 2^-31 = (#times r was a positive) - (#times r was a negative) - (what was overwritten); the triplets of the epoch come
from gorse_bpr_sample_triplets.  |expected - got| summed over the cold rows is a lower bound of the updates lost there
(a lost positive and a lost negative of one row cancel), reported relative to the updates those rows received.

usage: gpu_probe_bpr_stores.py [c2] [c3s] [c3]      Output -> profiles/rNN_*_probe_bpr_stores.txt"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the positive-side and re-reading forms of the store route live in the probe build only (make -C gorse_amd/csrc probe-lib)
_probe = os.path.join(ROOT, "gorse_amd", "lib", "libgorse_hip_probe.so")
if os.path.exists(_probe) and os.path.getmtime(_probe) >= os.path.getmtime(os.path.join(ROOT, "gorse_amd", "lib", "libgorse_hip.so")):
    os.environ.setdefault("GORSE_HIP_LIB", _probe)
import numpy as np

from gorse_amd import capi, synth
from oracle import oracle as orc

L = capi.lib()
o = orc.Oracle()
args = sys.argv[1:] or ["c2", "c3s"]
UNIT = 2.0 ** -31


def lost_updates(mf, data, d, seed, epoch, n):
    """(updates to cold rows, lower bound of the lost ones, updates to all rows, lost on all rows) for one epoch of n samples"""
    U, I = data.U, data.I
    P = np.full((U, d), 2.0 ** -10, np.float32)
    Q = np.zeros((I, d), np.float32)
    mf.set_factors(P, Q)
    mf.bpr_epoch(n, 2.0 ** -20, 0.0, seed, epoch)
    _, gq = mf.get_factors()
    u, i, j = mf.bpr_sample_triplets(n, seed, epoch)
    ok = u >= 0
    pos = np.bincount(i[ok], minlength=I).astype(np.int64)
    neg = np.bincount(j[ok], minlength=I).astype(np.int64)
    out = []
    for col in (0, d - 1):  # first and last 64-byte piece of the row
        got = np.rint(gq[:, col].astype(np.float64) / UNIT).astype(np.int64)
        out.append(np.abs((pos - neg) - got))
    miss = np.maximum(out[0], out[1])
    return pos + neg, miss


def run_case(name, data, d, epochs, ref_ndcg, variants, n_loss):
    U, I = data.U, data.I
    P0, Q0 = synth.init_factors(U, I, d, 0.0, 0.001, 1)
    share = np.bincount(data.uidx, minlength=I) / float(data.n_train)
    for label, window, store, extra_variant in variants:
        L.gorse_hip_test_set_bpr_cold_window(window)
        L.gorse_hip_test_set_bpr_store_mode(store)
        L.gorse_hip_test_set_variant(128 | extra_variant)
        mf = capi.MF(U, I, d, data.uptr, data.uidx)
        cold = (share + 1.0 / I) * window < 1.0 if window > 0 else np.zeros(I, bool)
        mf.set_factors(P0, Q0)
        mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 1, 9)  # warm-up: buffers, code objects
        mf.synchronize()
        mf.set_factors(P0, Q0)
        mf.set_profiling(True)
        mf.reset_profile()
        t0 = time.perf_counter()
        for e in range(epochs):
            mf.bpr_epoch_enqueue(data.n_train, 0.05, 0.01, 77, 1 + e)
        mf.synchronize()
        wall = time.perf_counter() - t0
        n, ms = mf.get_profile(capi.PROF_BPR_UPDATE)
        _, mss = mf.get_profile(capi.PROF_BPR_SORT)
        _, msq = mf.get_profile(capi.PROF_BPR_SAMPLE)
        mf.set_profiling(False)
        gp, gq = mf.get_factors()
        finite = bool(np.isfinite(gp).all() and np.isfinite(gq).all())
        ndcg = float("nan")
        if finite and data.test_idx.size:
            ndcg = o.evaluate(gp, gq, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0]
        lost = ""
        if store and n_loss:
            touched, miss = lost_updates(mf, data, d, 5, 3, n_loss)
            tc, mc = int(touched[cold].sum()), int(miss[cold].sum())
            tw, mw = int(touched[~cold].sum()), int(miss[~cold].sum())
            lost = " | cold rows %d (%.1f %% of items) got %d updates, lost >= %d (%.3f %%); other rows lost >= %d of %d" % (
                int(cold.sum()), 100.0 * cold.mean(), tc, mc, 100.0 * mc / max(tc, 1), mw, tw)
        print("%-4s d=%3d %-44s update %8.3f ms/epoch (%.3e samples/s) sort %.3f sampler %.3f wall %.3f ms/epoch (%.3e/s) "
              "NDCG %.4f (seq %.4f)%s" % (name, d, label, ms / epochs, epochs * data.n_train / (ms * 1e-3), mss / epochs, msq / epochs,
                                          wall / epochs * 1e3, epochs * data.n_train / wall, ndcg, ref_ndcg, lost), flush=True)
        mf.close()
    L.gorse_hip_test_set_bpr_cold_window(-1)
    L.gorse_hip_test_set_bpr_store_mode(-1)
    L.gorse_hip_test_set_variant(0)


def store_variants(windows):
    v = [("atomics only", 0, 0, 0), ("atomics only, round-3 preparation", 0, 0, 1 << 26)]
    for w in windows:
        v += [("W=%d stores: negatives" % w, w, 1, 0), ("W=%d stores: negatives + positives" % w, w, 3, 0),
              ("W=%d stores: negatives, live re-read" % w, w, 5, 0), ("W=%d stores: both, live re-read" % w, w, 7, 0)]
    v += [("stores on every item but the hot ones", 1, 3, 0)]
    return v


if "c2" in args:
    data = synth.s_ml1m()
    d, epochs = 64, 8
    srt = orc.sort_rows(data.uptr, data.uidx)
    P, Q = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 1)
    for ep in range(epochs):
        o.bpr_epoch_sampled(P, Q, data.uptr, data.uidx, srt, 77, 1 + ep, 0, data.n_train, 0.05, 0.01)
    ref = o.evaluate(P, Q, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0]
    run_case("c2", data, d, epochs, ref, store_variants([32768, 2048]), data.n_train)
if "c3s" in args:
    data = synth.hold_out(synth.s_big_shard(rank=0, world=8), 8192, 99, 5)
    d = 128
    srt = orc.sort_rows(data.uptr, data.uidx)
    P, Q = synth.init_factors(data.U, data.I, d, 0.0, 0.001, 1)
    t0 = time.perf_counter()
    o.bpr_epoch_sampled(P, Q, data.uptr, data.uidx, srt, 77, 1, 0, data.n_train, 0.05, 0.01)
    ref = o.evaluate(P, Q, data.test_ptr, data.test_idx, data.neg_ptr, data.neg_idx, 10)[0]
    print("c3s sequential oracle epoch: %.0f s" % (time.perf_counter() - t0), flush=True)
    run_case("c3s", data, d, 1, ref, store_variants([32768]), data.n_train)
if "c3" in args:
    gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c3full_oracle_ndcg.json")))
    t0 = time.perf_counter()
    data = synth.hold_out(synth.s_big_full(), 8192, 99, 5)
    print("c3 data set ready in %.0f s" % (time.perf_counter() - t0), flush=True)
    run_case("c3", data, 128, 1, gold["ndcg_after_one_epoch"], store_variants([32768, 8192]), 32_000_000)
