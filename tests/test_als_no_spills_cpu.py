"""No kernel of the ALS path spills (DESIGN.md section 4 "ALS", round 6).  Rounds 3-5 carried 34 .. 335 scalar spills in the kernels of
65 <= nFactors <= 128 and in the padded chunk kernels: one bound check per tile element, invariant across the kernel's row loop, whose
masks hipcc kept in scalar registers over the loop and -- out of those -- in vector lanes, with a `v_readlane` pair in front of every
store (nFactors 128 on C5: 25.8 -> 22.75 ms per epoch once they were gone).  The compiler honours no "do not hoist" request, so the state is
held by construction -- and checked here, on the gfx950 assembly hipcc emits for als.hip (no device needed)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_als_kernel_spills():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_census.py"), os.path.join(ROOT, "gorse_amd", "csrc", "als.hip")],
                         capture_output=True, text=True, check=True).stdout
    seen = {}
    for line in out.splitlines():
        m = re.match(r"^(?:void )?(\S.*?)\s+vgpr\s+(\d+)\s+agpr\s+(\d+)\s+sgpr\s+(\d+)\s+spills: vgpr (\d+) sgpr (\d+)\s+scratch (\d+) B", line)
        if m:
            seen[m.group(1)] = tuple(int(m.group(i)) for i in (2, 5, 6, 7))
    # what half_epoch can launch for nFactors <= 128, and the probes' comparison forms beside them
    for name in ("als_wide_kernel<false, 2>", "als_wide_kernel<true, 2>", "als_wide_kernel<false, 1>", "als_wide_kernel<true, 1>",
                 "als_wide_kernel<false, 0>", "als_wide_long_kernel", "als_gram_partial_kernel", "als_chunk_kernel<4, 3>",
                 "als_chunk_kernel<4, 4>", "als_row_kernel<2, 2, 8>", "als_long_solve_kernel<64>"):
        assert name in seen, (name, sorted(seen))
    assert len(seen) >= 45
    for name, (vgpr, vspill, sspill, scratch) in seen.items():
        assert vspill == 0 and scratch == 0, (name, vspill, scratch)
        # (the wide row kernel's prologue parks a handful of kernel arguments in vector lanes, once per workgroup: not a per-row cost)
        assert sspill <= (8 if name.startswith("als_wide_kernel<false") else 0), (name, sspill)
        assert vgpr <= 256, (name, vgpr)
