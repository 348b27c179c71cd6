#!/usr/bin/env python3
"""C5's ITEM half-sweep at full size (500K x 100K x 50M, nFactors 64): device (Gram form: chunk plan + partial reduce + long solve for the
rows of more than 4096 entries) against the oracle (the reference's float32 residual recurrence, model.go:707-738) and against the same
recurrence in float64, per row length.  Which of the two float32 answers is the one that drifts on a 4-million-entry row?"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gorse_amd import capi, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def half_fp64(A, B, ptr, idx, S, w, reg, rows):
    out = []
    for u in rows:
        fb = idx[ptr[u]:ptr[u + 1]]
        Bu = B[fb].astype(np.float64)
        pu = A[u].astype(np.float64).copy()
        pred = Bu @ pu
        for f in range(A.shape[1]):
            q = Bu[:, f]
            res = pred - pu[f] * q
            a = ((1 - (1 - w) * res) * q).sum()
            c = ((1 - w) * q * q).sum()
            b = w * (pu @ S[:, f] - pu[f] * S[f, f])
            pu[f] = (a - b) / (c + w * S[f, f] + reg)
            pred = res + pu[f] * q
        out.append(pu)
    return np.array(out)


def main():
    U, I, d, w, reg = 500_000, 100_000, 64, 0.001, 0.06
    uptr, uidx, iptr, iidx = synth.s_als(U, I, 50_000_000, 45)
    P0, Q0 = synth.init_factors(U, I, d, 0.0, 0.1, seed=1)
    mf = capi.MF(U, I, d, uptr, uidx, iptr, iidx)
    mf.set_factors(P0, Q0)
    mf.als_half_epoch(1, w, reg)
    _, gQ = mf.get_factors()
    lens = np.diff(iptr)
    by_len = np.argsort(lens, kind="stable")
    rows = np.unique(np.concatenate([by_len[np.linspace(0, I - 1, 200).astype(np.int64)], by_len[-40:]]))
    rows = rows[np.argsort(lens[rows])]
    o = orc.Oracle()
    A = np.ascontiguousarray(Q0[rows])
    sub_ptr = np.zeros(rows.size + 1, np.int64)
    np.cumsum(lens[rows], out=sub_ptr[1:])
    sub_idx = np.concatenate([iidx[iptr[r]:iptr[r + 1]] for r in rows])
    t0 = time.perf_counter()
    o.als_half_range(A, P0, sub_ptr, sub_idx, uptr, w, reg, 0, rows.size)
    t_orc = time.perf_counter() - t0
    has = np.diff(uptr) > 0
    S = P0[has].astype(np.float64).T @ P0[has].astype(np.float64)
    t0 = time.perf_counter()
    X = half_fp64(Q0, P0, iptr, iidx, S, w, reg, rows)
    print("oracle %.1f s, float64 %.1f s for %d rows" % (t_orc, time.perf_counter() - t0, rows.size), flush=True)
    scale = np.abs(X).max(axis=1)
    e_dev = np.abs(gQ[rows] - X).max(axis=1) / scale
    e_orc = np.abs(A - X).max(axis=1) / scale
    e_do = np.abs(gQ[rows].astype(np.float64) - A).max(axis=1) / scale
    print("%10s %12s %12s %12s" % ("entries", "dev-f64", "oracle-f64", "dev-oracle"))
    edges = [0, 64, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 1 << 23]
    for lo, hi in zip(edges, edges[1:]):
        m = (lens[rows] > lo) & (lens[rows] <= hi)
        if m.any():
            print("(%7d, %7d]: %3d rows  device-f64 max %.2e  oracle-f64 max %.2e  device-oracle max %.2e"
                  % (lo, hi, int(m.sum()), e_dev[m].max(), e_orc[m].max(), e_do[m].max()))
    for t in range(rows.size - 12, rows.size):
        print("%10d %12.2e %12.2e %12.2e" % (lens[rows[t]], e_dev[t], e_orc[t], e_do[t]))


if __name__ == "__main__":
    main()
