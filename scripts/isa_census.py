#!/usr/bin/env python3
"""Registers and spills per kernel of one .hip file: compiles it to gfx950 assembly (device only) and reads the kernel metadata.
usage: isa_census.py gorse_amd/csrc/als.hip [name substring] [-DGORSE_PROBE]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
pat = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
extra = [a for a in sys.argv[2:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as t:
    out = os.path.join(t, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics",
                    "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-x", "hip", src, "-o", out] + extra,
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
demangle = subprocess.run(["c++filt"], input="\n".join(re.findall(r"^\s+\.name:\s+(\S+)", text, re.M)),
                          capture_output=True, text=True).stdout.split("\n")
names = dict(zip(re.findall(r"^\s+\.name:\s+(\S+)", text, re.M), demangle))
for block in text.split("  - .agpr_count:")[1:]:
    block = ".agpr_count:" + block
    f = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", block)}
    name = names.get(f.get("name", ""), f.get("name", ""))
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    if pat in name:
        print("%-70s vgpr %3s agpr %3s sgpr %3s  spills: vgpr %s sgpr %s  scratch %s B  lds %s" % (
            name[:70], f.get("vgpr_count"), f.get("agpr_count"), f.get("sgpr_count"), f.get("vgpr_spill_count"), f.get("sgpr_spill_count"),
            f.get("private_segment_fixed_size"), f.get("group_segment_fixed_size")))
