#!/usr/bin/env python3
"""GPU probe: one ALS epoch (model/cf/model.go:641-738) on a C5-shaped input, residual sweep vs Gram form.
Prints one line per (shape, path): epoch wall time, hipEvent time of the sweeps / Gram kernels, the algorithmic
gather rate 2*nnz*d*4 + 2*(U+I)*d*4 bytes per epoch and the agreement between the two paths."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gorse_amd import capi, synth  # noqa: E402


def run(name, U, I, nnz, d, paths, reps=3):
    t0 = time.perf_counter()
    uptr, uidx, iptr, iidx = synth.s_als(U, I, nnz, 45)
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
        import zlib
        print("%-26s path %d factors crc32 P %08x Q %08x" % (name, path, zlib.crc32(gP.tobytes()), zlib.crc32(gQ.tobytes())), flush=True)
        algo = 2.0 * n * d * 4 + 2.0 * (U + I) * d * 4
        print("%-26s path %d U=%7d I=%7d nnz=%9d d=%3d epoch %9.3f ms (sweeps %9.3f ms, S-gram %7.3f ms) algorithmic %6.1f GB/s "
              "max item row %d finite=%s gen %.1fs" % (name, path, U, I, n, d, dt * 1e3, sweep / reps, gram / reps, algo / dt / 1e9,
                                                      int(np.diff(iptr).max()), bool(np.isfinite(gP).all() and np.isfinite(gQ).all()), gen),
              flush=True)
        del mf
    if len(res) >= 2:
        (aP, aQ), (bP, bQ) = list(res.values())[0], list(res.values())[-1]
        scale = max(np.abs(aP).max(), np.abs(aQ).max())
        print("%-26s paths agree to %.2e (P) %.2e (Q) of the factor scale after %d epochs"
              % (name, np.abs(aP - bP).max() / scale, np.abs(aQ - bQ).max() / scale, reps + 1), flush=True)
    capi.lib().gorse_hip_test_set_als_path(0)


def prof(name, U, I, nnz, d, path=0):
    """phase counters of als_row_kernel (s_memtime) on one epoch"""
    capi.lib().gorse_hip_test_set_als_path(path)
    uptr, uidx, iptr, iidx = synth.s_als(U, I, nnz, 45)
    P, Q = synth.init_factors(U, I, d, 0.0, 0.1, 1)
    mf = capi.MF(U, I, d, uptr, uidx, iptr, iidx)
    mf.set_factors(P, Q)
    mf.als_epoch(0.001, 0.06)
    mf.als_profile(True)
    t0 = time.perf_counter()
    mf.als_epoch(0.001, 0.06)
    dt = time.perf_counter() - t0
    c = mf.als_profile(False, fetch=True)
    mf.als_epoch(0.001, 0.06)
    t0 = time.perf_counter()
    for _ in range(3):
        mf.als_epoch(0.001, 0.06)
    plain = (time.perf_counter() - t0) / 3
    print("%-18s epoch without stamps %.2f ms" % (name, plain * 1e3), flush=True)
    for side, label in ((0, "user rows"), (1, "item rows (short)")):
        a, m, s, rows, ent, tot, waves, ld = c[8 * side:8 * side + 8]
        waves = max(waves, 1)
        print("%-18s %-18s rows %8d entries %10d | per wave: kernel %.3e ticks = accumulate %.1f%% + M to LDS %.1f%% + solve %.1f%% "
              "| per row: accumulate %.0f, M %.0f, solve %.0f ticks (of which the columns of M into registers %.0f); %.1f entries per row (epoch with stamps %.2f ms)"
              % (name, label, rows, ent, tot / waves, 100.0 * a / max(tot, 1), 100.0 * m / max(tot, 1), 100.0 * s / max(tot, 1),
                 a / max(rows, 1), m / max(rows, 1), s / max(rows, 1), ld / max(rows, 1), ent / max(rows, 1), dt * 1e3), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "wide":  # nFactors 128: als_wide_kernel with G by fused multiply-adds (8, round 2) against the fp32 MFMA (0)
        # 8: G by fused multiply-adds (round 2); 1024: the fp32 MFMA (round 3); 0: the bf16 MFMA over three-way split values (round 4)
        run("20Kx10Kx1M d=128", 20_000, 10_000, 1_000_000, 128, (8, 1024, 0), reps=2)
        run("C5 shard/4 d=128", 125_000, 100_000, 12_500_000, 128, (8, 1024, 0, 1024, 0), reps=3)
        run("C5 shard/4 d=96", 125_000, 100_000, 12_500_000, 96, (8, 1024, 0), reps=2)
        prof("C5 shard/4 d=128, fp32 MFMA", 125_000, 100_000, 12_500_000, 128, 1024)
        prof("C5 shard/4 d=128, bf16 x 3", 125_000, 100_000, 12_500_000, 128)
        capi.lib().gorse_hip_test_set_als_path(0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "spills":  # round 6: the kernels that lost their scalar spills (A/B against the build before)
        run("C5 full d=128", 500_000, 100_000, 50_000_000, 128, (0,), reps=3)
        run("C5 shard/4 d=96", 125_000, 100_000, 12_500_000, 96, (0,), reps=3)
        run("C5 shard/4 d=56", 125_000, 100_000, 12_500_000, 56, (0,), reps=3)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "phased":  # accumulate / solve in lockstep per workgroup (path | 4) against free-running
        prof("C5 d=64 free-running", 500_000, 100_000, 50_000_000, 64, 0)
        prof("C5 d=64 phased", 500_000, 100_000, 50_000_000, 64, 4)
        run("C5 full d=64", 500_000, 100_000, 50_000_000, 64, (0, 4), reps=3)
        capi.lib().gorse_hip_test_set_als_path(0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "tiles":  # round 4: 16 x 16 MFMA tiles (0; 256: at 8 waves per workgroup) against 32 x 32 (128)
        # 128 | 1024: fp32 32 x 32 tiles; 1024: fp32 16 x 16 tiles where the default is the bf16 x 3 form; 256: 8 waves per workgroup
        for d in (64, 48, 32, 16):
            run("C5 shard/4 d=%d" % d, 125_000, 100_000, 12_500_000, d, (128 | 1024, 1024, 256, 0), reps=3)
        run("C5 full d=64", 500_000, 100_000, 50_000_000, 64, (128 | 1024, 1024, 0, 1024, 0), reps=8)
        prof("C5 d=64, 32x32 tiles", 500_000, 100_000, 50_000_000, 64, 128 | 1024)
        prof("C5 d=64, 16x16 tiles", 500_000, 100_000, 50_000_000, 64, 1024)
        prof("C5 d=64, bf16 x 3", 500_000, 100_000, 50_000_000, 64, 0)
        capi.lib().gorse_hip_test_set_als_path(0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "prof":
        prof("C5 d=64", 500_000, 100_000, 50_000_000, 64)
        prof("C5 shard d=16", 125_000, 100_000, 12_500_000, 16)
        return
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    run("small 20Kx10Kx1M", 20_000, 10_000, 1_000_000, 64, (1, 2))
    run("C5 shard/4 d=64", 125_000, 100_000, 12_500_000, 64, (2,) if quick else (1, 2), reps=2)
    run("C5 shard/4 d=16", 125_000, 100_000, 12_500_000, 16, (2,), reps=2)
    if not quick:
        run("C5 full d=64", 500_000, 100_000, 50_000_000, 64, (2,), reps=2)


if __name__ == "__main__":
    main()
