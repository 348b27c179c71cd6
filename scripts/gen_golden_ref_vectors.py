#!/usr/bin/env python3
"""Generate tests/golden/ref_simd_vectors.npz from the REFERENCE's own C kernels.

Runs only in the authoring container (needs /root/reference and an AVX512 host):
oracle/Makefile `ref` compiles common/floats/src/floats_avx{,512}.c and
common/bfloats/src/bfloats_avx{,512}.c where they lie, this script feeds them
seeded random vectors of every length 0..200 (so every body / 8-tail / scalar-tail
combination occurs) and stores inputs + outputs.  tests/test_oracle_golden.py then
requires the oracle restatement to reproduce every output BIT-EXACTLY, on any host
(the fixture travels; /root/reference does not).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

orc.build(force=False)
ref = orc.load_ref()
assert ref is not None, "needs oracle/_ref/libgorse_ref.so and an AVX512 host"

rng = np.random.default_rng(20260921)
lengths = list(range(0, 201)) + [256, 512, 1000]
out = {"lengths": np.array(lengths, dtype=np.int64)}
A, B, Cv = [], [], []
res = {k: [] for k in ("dot512", "dot256", "euc512", "euc256", "bfeuc512", "bfeuc256")}
mca512, mca256, mcat512 = [], [], []
scal = []
for n in lengths:
    a = rng.standard_normal(n).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    c = rng.standard_normal(n).astype(np.float32)
    s = np.float32(rng.standard_normal())
    A.append(a); B.append(b); Cv.append(c); scal.append(s)
    if n > 0:
        res["dot512"].append(ref.dot(orc.ISA_AVX512, a, b))
        res["dot256"].append(ref.dot(orc.ISA_AVX, a, b))
        res["euc512"].append(ref.euclidean(orc.ISA_AVX512, a, b))
        res["euc256"].append(ref.euclidean(orc.ISA_AVX, a, b))
        ab = (a.view(np.uint32) >> 16).astype(np.uint16)
        bb = (b.view(np.uint32) >> 16).astype(np.uint16)
        res["bfeuc512"].append(ref.euclidean_bf16(orc.ISA_AVX512, ab, bb))
        res["bfeuc256"].append(ref.euclidean_bf16(orc.ISA_AVX, ab, bb))
    else:  # the Go wrappers never call the kernels with n == 0
        for k in res:
            res[k].append(0.0)
    mca512.append(ref.mul_const_add(orc.ISA_AVX512, a, float(s), c))
    mca256.append(ref.mul_const_add(orc.ISA_AVX, a, float(s), c))
    mcat512.append(ref.mul_const_add_to(orc.ISA_AVX512, a, float(s), c))
out["a"] = np.concatenate(A); out["b"] = np.concatenate(B); out["c"] = np.concatenate(Cv)
out["s"] = np.array(scal, dtype=np.float32)
for k, v in res.items():
    out[k] = np.array(v, dtype=np.float32)
out["mca512"] = np.concatenate(mca512); out["mca256"] = np.concatenate(mca256)
out["mcat512"] = np.concatenate(mcat512)

# GEMM: all four transpose cases, sizes that exercise 16-body/8-tail/scalar-tail columns
mm_cases = []
for (m, n, k) in [(3, 5, 7), (4, 16, 16), (5, 24, 9), (2, 37, 33), (7, 64, 48)]:
    for tA in (0, 1):
        for tB in (0, 1):
            a = rng.standard_normal((k, m) if tA else (m, k)).astype(np.float32)
            b = rng.standard_normal((n, k) if tB else (k, n)).astype(np.float32)
            c0 = rng.standard_normal((m, n)).astype(np.float32)
            lda, ldb = a.shape[1], b.shape[1]
            c512 = ref.mm(orc.ISA_AVX512, tA, tB, m, n, k, a.ravel(), lda, b.ravel(), ldb, c0.ravel(), n)
            c256 = ref.mm(orc.ISA_AVX, tA, tB, m, n, k, a.ravel(), lda, b.ravel(), ldb, c0.ravel(), n)
            mm_cases.append((m, n, k, tA, tB, a.ravel(), b.ravel(), c0.ravel(), c512, c256))
out["mm_shapes"] = np.array([[c[0], c[1], c[2], c[3], c[4]] for c in mm_cases], dtype=np.int64)
out["mm_a"] = np.concatenate([c[5] for c in mm_cases]); out["mm_b"] = np.concatenate([c[6] for c in mm_cases])
out["mm_c0"] = np.concatenate([c[7] for c in mm_cases])
out["mm_c512"] = np.concatenate([c[8] for c in mm_cases]); out["mm_c256"] = np.concatenate([c[9] for c in mm_cases])

path = os.path.join(ROOT, "tests", "golden", "ref_simd_vectors.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
