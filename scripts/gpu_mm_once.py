#!/usr/bin/env python3
"""floats.MM 4096^3 NN, five calls: the command the MM counter passes of scripts/gpu_session.sh profile (rocprofv3 --pmc ... -- python scripts/gpu_mm_once.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gorse_amd import capi

n = 4096
rng = np.random.default_rng(3)
a = rng.standard_normal((n, n)).astype(np.float32)
b = rng.standard_normal((n, n)).astype(np.float32)
c = np.zeros((n, n), np.float32)
for _ in range(5):
    capi.sgemm(0, 0, n, n, n, a.ravel(), n, b.ravel(), n, c.ravel(), n)
    print("kernel %.3f ms" % capi.lib().gorse_hip_test_sgemm_last_ms(), flush=True)
