#!/usr/bin/env python3
"""GPU probe: one ALS epoch (model/cf/model.go:641-738) on a C5-shaped input, residual sweep vs Gram form.
Prints one line per (shape, path): epoch wall time, hipEvent time of the sweeps / Gram kernels, the algorithmic
gather rate 2*nnz*d*4 + 2*(U+I)*d*4 bytes per epoch and the agreement between the two paths."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402


def fast_cf(U, I, nnz, seed, zipf=1.0):
    """log-normal user activity, Zipf item popularity, duplicates inside a row allowed (timing only)"""
    rng = np.random.default_rng(seed)
    act = rng.lognormal(0.0, 1.0, U)
    lens = np.maximum(1, np.floor(act / act.sum() * nnz)).astype(np.int64)
    uptr = np.zeros(U + 1, np.int64)
    np.cumsum(lens, out=uptr[1:])
    n = int(uptr[-1])
    w = 1.0 / np.power(np.arange(1, I + 1, dtype=np.float64), zipf)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    perm = rng.permutation(I).astype(np.int32)
    uidx = perm[np.minimum(np.searchsorted(cdf, rng.random(n)), I - 1)]
    rows = np.repeat(np.arange(U, dtype=np.int32), lens)
    order = np.argsort(uidx, kind="stable")
    iidx = rows[order]
    iptr = np.zeros(I + 1, np.int64)
    np.cumsum(np.bincount(uidx, minlength=I), out=iptr[1:])
    return uptr, uidx.astype(np.int32), iptr, iidx.astype(np.int32)


def run(name, U, I, nnz, d, paths, reps=3):
    t0 = time.perf_counter()
    uptr, uidx, iptr, iidx = fast_cf(U, I, nnz, 45)
    gen = time.perf_counter() - t0
    n = int(uptr[-1])
    P, Q = synth.init_factors(U, I, d, 0.0, 0.1, 1)
    res = {}
    for path in paths:
        capi.lib().gorse_hip_test_set_als_path(path)
        mf = capi.MF(U, I, d, uptr, uidx, iptr, iidx)
        mf.set_factors(P, Q)
        mf.als_epoch(0.001, 0.06)  # warm-up: allocations, code objects
        mf.set_factors(P, Q)
        mf.set_profiling(True)
        mf.reset_profile()
        t0 = time.perf_counter()
        for _ in range(reps):
            mf.als_epoch(0.001, 0.06)
        dt = (time.perf_counter() - t0) / reps
        ns, sweep = mf.get_profile(capi.PROF_ALS_SWEEP)
        ng, gram = mf.get_profile(capi.PROF_ALS_GRAM)
        gP, gQ = mf.get_factors()
        res[path] = (gP, gQ)
        algo = 2.0 * n * d * 4 + 2.0 * (U + I) * d * 4
        print("%-26s path %d U=%7d I=%7d nnz=%9d d=%3d epoch %9.3f ms (sweeps %9.3f ms, S-gram %7.3f ms) algorithmic %6.1f GB/s "
              "max item row %d finite=%s gen %.1fs" % (name, path, U, I, n, d, dt * 1e3, sweep / reps, gram / reps, algo / dt / 1e9,
                                                      int(np.diff(iptr).max()), bool(np.isfinite(gP).all() and np.isfinite(gQ).all()), gen),
              flush=True)
        del mf
    if len(res) == 2:
        (aP, aQ), (bP, bQ) = res.values()
        scale = max(np.abs(aP).max(), np.abs(aQ).max())
        print("%-26s paths agree to %.2e (P) %.2e (Q) of the factor scale after %d epochs"
              % (name, np.abs(aP - bP).max() / scale, np.abs(aQ - bQ).max() / scale, reps + 1), flush=True)
    capi.lib().gorse_hip_test_set_als_path(0)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    run("small 20Kx10Kx1M", 20_000, 10_000, 1_000_000, 64, (1, 2))
    run("C5 shard/4 d=64", 125_000, 100_000, 12_500_000, 64, (2,) if quick else (1, 2), reps=2)
    run("C5 shard/4 d=16", 125_000, 100_000, 12_500_000, 16, (2,), reps=2)
    if not quick:
        run("C5 full d=64", 500_000, 100_000, 50_000_000, 64, (2,), reps=2)


if __name__ == "__main__":
    main()
