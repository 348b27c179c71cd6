#!/usr/bin/env python3
"""GPU probe: floats.MM on the fp32 MFMA (csrc/sgemm.hip) -- kernel TFLOP/s per shape and transposition, and bit-equality with the
vector-ALU chain (gorse_hip_test_set_sgemm_valu) on each.   Output -> profiles/rNN_*_probe_mm.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gorse_amd import capi

L = capi.lib()
rng = np.random.default_rng(3)
for (m, n, k) in ((4096, 4096, 4096), (2048, 2048, 2048), (1024, 1024, 1024), (8192, 512, 1024), (513, 1027, 255), (100000, 64, 64)):
    for ta, tb in ((0, 0), (1, 0), (1, 1)):
        a = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
        b = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
        c0 = rng.standard_normal((m, n)).astype(np.float32)
        lda, ldb = a.shape[1], b.shape[1]
        out = {}
        for valu in (0, 1):
            L.gorse_hip_test_set_sgemm_valu(valu)
            c = c0.copy()
            capi.sgemm(ta, tb, m, n, k, a.ravel(), lda, b.ravel(), ldb, c.ravel(), n)
            ms = []
            for _ in range(3):
                c = c0.copy()
                capi.sgemm(ta, tb, m, n, k, a.ravel(), lda, b.ravel(), ldb, c.ravel(), n)
                ms.append(L.gorse_hip_test_sgemm_last_ms())
            out[valu] = (c, float(np.median(ms)))
            if valu == 1 and m * n * k > 2 ** 33:
                break
        L.gorse_hip_test_set_sgemm_valu(0)
        same = np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)) if 1 in out else None
        t = out[0][1]
        print("%6d x %5d x %5d %s%s: MFMA %8.3f ms = %6.1f TFLOP/s; vector ALU %8.3f ms; bit-equal %s" % (
            m, n, k, "T" if ta else "N", "T" if tb else "N", t, 2.0 * m * n * k / (t * 1e-3) / 1e12, out[1][1] if 1 in out else float("nan"), same), flush=True)
