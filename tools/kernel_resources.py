#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel of one .hip source, compiled for gfx950 with build.py's flags (device side
only; from the code object's metadata).
usage: tools/kernel_resources.py pcpx_query.hip [substring of the demangled name] [-DNAME=VALUE ...]"""
import importlib, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("point-cloud-processing_amd.build")
src = os.path.join(b.CSRC, sys.argv[1])
pat = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
extra = [a for a in sys.argv[2:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "dev.s")
    subprocess.check_call([b._hipcc()] + b.FLAGS + extra + ["--cuda-device-only", "-S", src, "-o", asm])
    txt = open(asm).read()
    txt = txt[txt.rfind("amdhsa.kernels:"):]
cur, rows = {}, []
for line in txt.split("\n"):
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "symbol"): cur[k] = v
    if k == "wavefront_size":
        rows.append(cur); cur = {}
syms = [r.get("symbol", "?").replace(".kd", "") for r in rows]
dem = subprocess.check_output(["c++filt"], input="\n".join(syms), text=True).split("\n")
for r, nm in zip(rows, dem):
    nm = re.sub(r"pcpx::\(anonymous namespace\)::|void |pcpx::", "", nm)
    if pat and pat not in nm: continue
    print("%-72s vgpr %3s sgpr %3s sspill %3s vspill %3s scratch %4s lds %6s" % (nm[:72], r.get("vgpr_count"), r.get("sgpr_count"), r.get("sgpr_spill_count"), r.get("vgpr_spill_count"), r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size")))
