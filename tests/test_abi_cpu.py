"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/gorse_hip.h declares (and nothing in the product imports the oracle), argument validation
that needs no GPU behaves like the reference's error paths."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(names=("gorse_hip.h", "gorse_hip_test.h")):
    out = set()
    for name in names:
        src = open(os.path.join(ROOT, "include", name)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(gorse_[a-z0-9_]+)\s*\(", src))
    return sorted(out)


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from gorse_amd import capi
    L = C.CDLL(capi.LIB_PATH)
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "libgorse_hip.so does not export %s" % n
    # the ctypes table mirrors the header one to one
    assert set(capi.SIGNATURES) == set(names)
    assert capi.lib().gorse_hip_abi_version() == 1
    # the drop-in boundary carries no test hook, the hook header nothing else
    assert not [n for n in header_functions(("gorse_hip.h",)) if n.startswith("gorse_hip_test_")]
    assert all(n.startswith("gorse_hip_test_") for n in header_functions(("gorse_hip_test.h",)))


def test_no_device_fails_loudly():
    """Without a GPU every compute entry point must fail with an error, never fall back."""
    from gorse_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.GorseHipError) as e:
        capi.MF(4, 4, 8, np.array([0, 1, 2, 3, 4], np.int64), np.array([0, 1, 2, 3], np.int32))
    assert e.value.code == capi.ERR_NO_DEVICE
    with pytest.raises(capi.GorseHipError) as e:
        capi.TopK(np.ones((4, 4), np.float32), capi.METRIC_NEG_DOT)
    assert e.value.code == capi.ERR_NO_DEVICE
    with pytest.raises(capi.GorseHipError) as e:
        capi.sgemm(0, 0, 2, 2, 2, np.zeros(4), 2, np.zeros(4), 2, np.zeros(4), 2)
    assert e.value.code == capi.ERR_NO_DEVICE
    with pytest.raises(capi.GorseHipError) as e:
        capi.Sparse(np.array([0, 2], np.int64), np.array([1, 2], np.uint32), np.ones(2, np.float32))
    assert e.value.code == capi.ERR_NO_DEVICE
    # the host twins above the ABI have no CPU path either: a sparse query on hip:// ends in the library's error
    from gorse_amd import vectors as V
    db = V.Open("hip://")
    db.AddCollection("s", 0, V.Dot)
    db.AddVectors("s", [V.Vector("a", [1.0], Indices=[3])])
    with pytest.raises(RuntimeError, match="no HIP device"):
        db.QueryVectors("s", V.Vector(Values=[1.0], Indices=[3]), None, 1)


def test_argument_validation_without_gpu():
    from gorse_amd import capi
    with pytest.raises(capi.GorseHipError) as e:  # nFactors <= 0
        capi.MF(4, 4, 0, np.zeros(5, np.int64), np.zeros(0, np.int32))
    assert e.value.code == capi.ERR_INVALID
    with pytest.raises(capi.GorseHipError) as e:  # item index out of range
        capi.MF(2, 2, 8, np.array([0, 1, 2], np.int64), np.array([0, 7], np.int32))
    assert e.value.code == capi.ERR_INVALID
    with pytest.raises(capi.GorseHipError) as e:  # sparse rows must come with strictly ascending indices
        capi.Sparse(np.array([0, 2], np.int64), np.array([2, 2], np.uint32), np.ones(2, np.float32))
    assert e.value.code == capi.ERR_INVALID
    with pytest.raises(capi.GorseHipError) as e:  # leading dimension too small (floats.MM panics)
        capi.sgemm(0, 0, 2, 2, 2, np.zeros(4), 1, np.zeros(4), 2, np.zeros(4), 2)
    assert e.value.code == capi.ERR_INVALID
    assert b"leading dimension" in capi.lib().gorse_hip_last_error()


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under gorse_amd/ or include/ may reference it."""
    bad = []
    for base in ("gorse_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            if "lib" in dp.split(os.sep)[-2:]:
                continue
            for f in fs:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".c")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"(from|import)\s+oracle|liboracle|orc_[a-z_]+\s*\(|oracle/_ref", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
