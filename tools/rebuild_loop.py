#!/usr/bin/env python3
"""N index rebuilds of a device-resident cloud (for rocprofv3 --kernel-trace --stats): python tools/rebuild_loop.py [n] [reps] [kind]"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("point-cloud-processing_amd")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kind = sys.argv[3] if len(sys.argv) > 3 else "uniform"
coarse = len(sys.argv) > 4 and sys.argv[4] == "coarse"  # PCPX_BUILD_COARSE_ORDER
pts = pkg.synthetic.uniform_cloud(n, 43) if kind == "uniform" else pkg.synthetic.clustered_cloud(n, 44)
d = torch.from_numpy(pts).cuda()
torch.cuda.synchronize()
ix = pkg.Index.from_device(d.data_ptr(), n)
ix.rebuild_dev(d.data_ptr(), n, coarse_order=coarse)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ix.rebuild_dev(d.data_ptr(), n, coarse_order=coarse)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / reps
print(json.dumps({"n": n, "kind": kind, "coarse_order": coarse, "rebuild_ms": round(ms, 4), "bytes_per_point_algorithmic": 30, "GBps_algorithmic": round(30 * n / ms / 1e6, 1),
                  "frac_of_8TBps": round(30 * n / (ms * 1e-3) / 8e12, 4)}))
