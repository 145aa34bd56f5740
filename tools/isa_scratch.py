#!/usr/bin/env python3
"""Static spill and scratch counts per kernel from `make -C ray-tracing_amd/csrc asm`.

    python tools/isa_scratch.py [build dir] > profiles/r<round>_isa_scratch.txt

Counts, per kernel symbol of the gfx950 .s file, the vector and scalar instructions, the
scratch loads / stores (vector registers spilled to scratch memory) and the v_writelane /
v_readlane (scalar registers spilled into lanes of a vector register), and quotes the
resource remarks (-Rpass-analysis=kernel-resource-usage) of the shading and screen kernels.
"""
import glob
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
build = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "ray-tracing_amd", "csrc", "build")
asm = [f for f in glob.glob(os.path.join(build, "*gfx950*.s"))]
if not asm:
    sys.exit("no gfx950 .s file under %s: run make asm first" % build)
print("# Static instruction counts per kernel in the gfx950 ISA of librtx_amd.so (make -C ray-tracing_amd/csrc asm):")
print("# scratch_load / scratch_store = vector registers spilled to and reloaded from scratch memory;")
print("# v_writelane / v_readlane = scalar registers spilled into lanes of a vector register.")
print("# Resource lines from -Rpass-analysis=kernel-resource-usage.")
name, counts = None, None
for line in open(asm[0]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, counts = m.group(1), dict(v=0, s=0, ld=0, st=0, wl=0, rl=0)
        continue
    if name is None:
        continue
    t = line.strip()
    if t.startswith(".Lfunc_end"):
        short = re.sub(r"(v)?(6DFrame|P|m).*$", r"\1", name) if "ILj" in name or "ILb" in name else re.sub(r"6DFrame.*$|P.*$", "", name)
        if "kernel" in name and "selftest" not in name:
            print("%-40s VALU %5d SALU %5d scratch_load %4d scratch_store %4d v_writelane %4d v_readlane %4d" %
                  (short, counts["v"], counts["s"], counts["ld"], counts["st"], counts["wl"], counts["rl"]))
        name = None
        continue
    if not t or t[0] in ";.":
        continue
    op = t.split()[0]
    if op.startswith("scratch_load"):
        counts["ld"] += 1
    elif op.startswith("scratch_store"):
        counts["st"] += 1
    elif op.startswith("v_writelane"):
        counts["wl"] += 1
    elif op.startswith("v_readlane"):
        counts["rl"] += 1
    if op.startswith("v_"):
        counts["v"] += 1
    elif op.startswith("s_"):
        counts["s"] += 1
print()
usage = open(os.path.join(build, "resource_usage.txt")).read().split("Function Name: ")[1:]
for block in usage:
    sym = block.split()[0]
    if not (sym.startswith("_Z13screen_kernel") or sym.startswith("_Z12shade_kernel")):
        continue
    fields = {}
    for key in ("VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"):
        m = re.search(r"    " + re.escape(key) + r": (\d+)", block)
        if m:
            fields[key] = m.group(1)
    print(re.sub(r"6DFrame.*$", "", sym), fields)
