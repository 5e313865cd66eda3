#!/usr/bin/env python3
"""Single-query latency, range-search latency and construction time in the shapes of the reference's own benchmarks
(benchmark/spatial_data_structures_benchmark.cpp:108-148, :169-213, :243-264):
tools/latency_bench.cpp through the drop-in C++ header on the GPU, the oracle's octree restatement (reference algorithm and
defaults, one thread, one query at a time) on the host beside it.  usage: python tools/latency_report.py [out.json]"""
import importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
PKG = os.path.join(ROOT, "point-cloud-processing_amd")
importlib.import_module("point-cloud-processing_amd.build").build()
exe = "/tmp/latency_bench"
subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "latency_bench.cpp"), "-o", exe,
                "-L", PKG, "-lpcpx", "-Wl,-rpath," + PKG, "-Wl,-rpath-link,/opt/rocm/lib", "-pthread"], check=True)
from oracle import pcp_oracle as O
out = {"shape": "one random query per iteration, k = 10, points U(-100,100)^3, voxel grid [-100,100]^3", "results": []}
for n in (1 << 20, 1 << 24):
    r = subprocess.run([exe, str(n), "3000"], capture_output=True, text=True, check=True)
    gpu = json.loads(r.stdout.strip().splitlines()[-1])
    rng = np.random.default_rng(5)
    m = n if n <= (1 << 22) else (1 << 22)  # the oracle's sequential insertion of 2^24 points takes minutes: the CPU side is timed up to 2^22
    pts = rng.uniform(-100, 100, (m, 3)).astype(np.float32)
    t0 = time.perf_counter(); tree = O.Octree(pts, voxel_grid=[-100, -100, -100, 100, 100, 100]); build_s = time.perf_counter() - t0
    q = rng.uniform(-100, 100, (3000, 3)).astype(np.float32)
    tree.knn(q[:100], 10, nthreads=1)
    t0 = time.perf_counter(); tree.knn(q, 10, nthreads=1); cpu_us = (time.perf_counter() - t0) / len(q) * 1e6
    # the reference's range-search shape on the host: one box of half-width <= 1 per call (benchmark :169-213)
    c = rng.uniform(-99, 99, (3000, 3)).astype(np.float32)
    blo = c + rng.uniform(-1, 0, (3000, 3)).astype(np.float32)
    bhi = c + rng.uniform(0, 1, (3000, 3)).astype(np.float32)
    t0 = time.perf_counter()
    for i in range(3000):
        tree.range_aabb(blo[i], bhi[i])
    gpu["cpu_oracle_octree_range_us_per_query_1_thread"] = round((time.perf_counter() - t0) / 3000 * 1e6, 2)
    t0 = time.perf_counter(); kd = O.KdTree(pts, compute_max_depth=True); gpu["cpu_oracle_kdtree_build_s"] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    for i in range(3000):
        kd.range_aabb(blo[i], bhi[i])
    gpu["cpu_oracle_kdtree_range_us_per_query_1_thread"] = round((time.perf_counter() - t0) / 3000 * 1e6, 2)
    gpu["cpu_oracle_octree_points"] = m
    gpu["cpu_oracle_octree_knn_us_per_query_1_thread"] = round(cpu_us, 2)
    gpu["cpu_oracle_octree_build_s"] = round(build_s, 2)
    out["results"].append(gpu)
    print(json.dumps(gpu), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
