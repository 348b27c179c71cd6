"""The library's own HOST code for the sparse top-k (gorse_amd/csrc/sparse.hip behind the C ABI: validation, index build on
the host and on the "device", uploads, scratch and stamp management, launches, downloads, statistics, error paths) in a
container without a GPU: a child pytest process runs tests/test_gpu_vectors_sparse.py -- the GPU parity tests themselves,
through gorse_amd.capi and the real libgorse_hip.so -- with tests/emu/libfakehip.so preloaded, a stand-in for the HIP runtime
entry points the library imports whose hipLaunchKernel runs the CPU emulation of csrc/sparse_kernels.hpp.  Test
infrastructure only: it shows that the host code and the kernel source agree with the oracle, nothing about gfx950."""
import os
import subprocess
import sys

import vectors_suite as S
from conftest import ROOT

EMU = os.path.join(ROOT, "tests", "emu")


def test_gpu_sparse_tests_through_the_real_library_on_a_fake_hip_runtime():
    so = os.path.join(EMU, "libfakehip.so")
    srcs = [os.path.join(EMU, f) for f in ("fake_hip.cpp", "fake_hip.map", "hip_emu.hpp")] + \
        [os.path.join(ROOT, "gorse_amd", "csrc", "sparse_kernels.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-pthread",
                               "-Wl,--version-script=" + srcs[1], "-o", so, srcs[0]])
    import __graft_entry__ as g
    g.build()  # libgorse_hip.so / libgorse_host.so up to date
    env = dict(os.environ, LD_PRELOAD=so, GORSE_GPU_ISOLATED="1")
    # the dense cases of that module need the MFMA / scan kernels, which are not emulated
    child = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                            os.path.join(ROOT, "tests", "test_gpu_vectors_sparse.py"), "-k",
                            " and ".join("not " + f.__name__ for f in S.PENDING_DENSE_CASES + [S.collaborative_recommend])],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    tail = (child.stdout + child.stderr)[-3000:]
    assert child.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail and "skipped" not in tail, tail
    assert int(tail.rsplit(" passed", 1)[0].split()[-1]) >= 24, tail


def test_bench_i2i_leg_on_the_fake_runtime():
    """bench.py --workload i2i (S-ml100k shape, single rank) end to end on the fake runtime: the JSON object carries the
    contract's fields, a roofline at 8 bytes per posting and the oracle baseline, whose rows are compared bit for bit with
    the library's inside bench.py"""
    so = os.path.join(EMU, "libfakehip.so")
    code = ("import json, sys, types; sys.argv = ['bench.py', '--workload', 'i2i', '--i2i-shape', 'ml100k', '--steps', '2', '--warmup', '1', "
            "'--cpu-seconds', '0.2']; import bench; args = bench.parse(); "
            "out = bench.bench_sparse(args, 1, 0, 0, lambda: None); print('JSON ' + json.dumps(out))")
    child = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, LD_PRELOAD=so), capture_output=True, text=True,
                           timeout=1200)
    assert child.returncode == 0, (child.stdout + child.stderr)[-3000:]
    import json
    out = json.loads([l for l in child.stdout.splitlines() if l.startswith("JSON ")][-1][5:])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["unit"] == "postings/s" and out["n_gpus"] == 1 and out["steps"] == 2 and out["vs_baseline"] is None
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] == "sparse_query_kernel" and r["algorithmic_bytes_per_posting"] == 8 and r["launches"] == 2
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert out["config"]["postings_per_step_per_gpu"] > 0 and out["config"]["queries_per_step_per_gpu"] == 1682
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and "compared bit for bit" in c["sample"]


def test_default_bench_keeps_its_line_whatever_the_sparse_leg_does():
    """bench.py appends the sparse leg to the default single-GPU line from a child process: no GPU here, so the child fails
    (or is killed at its time limit) and the parent must get an {"error": ...} object back, never an exception"""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    args = types.SimpleNamespace(i2i_shape="ml100k", cpu_seconds=0.1, no_cpu_baseline=True, i2i_timeout=300.0)
    out = bench.i2i_in_a_child(args)
    assert set(out) == {"error"} and "MI355X" in out["error"]  # "bench.py needs an MI355X (no CPU path exists)"
    args.i2i_timeout = 0.01
    out = bench.i2i_in_a_child(args)
    assert set(out) == {"error"} and "killed" in out["error"]
