#!/usr/bin/env python3
"""BPR.Fit / ALS.Fit of the C++ twin at the reference's own test shape (S-ml1m, 30 epochs, Verbose 10), second Fit of the process timed: wall
seconds, the log's fit_time / eval_time, and where the wall goes (for a kernel timeline).  usage: gpu_probe_fit.py <bpr | als> <nFactors>"""
import os
import re
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gorse_amd import cf, synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "bpr"
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
data = synth.s_ml1m()
train, test = cf.datasets_from_synth(data)
cfg = cf.NewFitConfig().SetJobs(max(2, os.cpu_count() or 2))


def make(seed=None):
    p = {"NFactors": d, "Reg": 0.01, "NEpochs": 30, "InitMean": 0, "InitStdDev": 0.001}
    if seed is not None:
        p["RandomState"] = seed
    if kind == "bpr":
        p["Lr"] = 0.05
        return cf.NewBPR(p)
    p = {"NFactors": d, "Reg": 0.015, "Alpha": 0.05, "NEpochs": 30}
    if seed is not None:
        p["RandomState"] = seed
    return cf.NewALS(p)


make().Fit(train, test, cfg)
for rep in range(3):
    m = make(1 + rep)
    t0 = time.perf_counter()
    score = m.Fit(train, test, cfg)
    wall = time.perf_counter() - t0
    fit_ms = [float(x) for x in re.findall(r"fit_time=([0-9.]+)ms", m.log)]
    eval_ms = [float(x) for x in re.findall(r"eval_time=([0-9.]+)ms", m.log)]
    print("%s.Fit nFactors %d: wall %.2f ms = epochs %.2f (30 x %.3f) + evaluations %.2f (%d x %.3f) + the rest %.2f; NDCG@10 %.4f"
          % (kind, d, wall * 1e3, 30 * np.mean(fit_ms), np.mean(fit_ms), np.sum(eval_ms), len(eval_ms), np.mean(eval_ms),
             wall * 1e3 - 30 * np.mean(fit_ms) - np.sum(eval_ms), score.NDCG), flush=True)
    print("   " + " | ".join(l for l in m.log.splitlines() if "setup" in l or "teardown" in l or " 0/" in l), flush=True)
