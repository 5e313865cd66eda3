#!/usr/bin/env python3
"""Build sort variants (tile size, staged scatter, look-back width) and time the rebuild with each (GPU box)."""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("point-cloud-processing_amd.build")
variants = {"i16_l8": [], "i16_l4": ["-DPCPX_SORT_LOOK=4"], "i16_l2": ["-DPCPX_SORT_LOOK=2"], "i24_l8": ["-DPCPX_SORT_ITEMS=24"],
            "i32_l8": ["-DPCPX_SORT_ITEMS=32"], "i32_l4": ["-DPCPX_SORT_ITEMS=32", "-DPCPX_SORT_LOOK=4"], "i24_l4": ["-DPCPX_SORT_ITEMS=24", "-DPCPX_SORT_LOOK=4"]}
for tag, flags in variants.items():
    lib = b.build(tag=tag, extra_flags=flags) if flags else b.build()
    env = dict(os.environ, PCPX_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rebuild_loop.py"), "1e7", "20"], capture_output=True, text=True, env=env)
    print(tag, r.stdout.strip() or r.stderr[-300:], flush=True)
