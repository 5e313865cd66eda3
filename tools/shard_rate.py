#!/usr/bin/env python3
"""Time of ONE rank's share of the bench step for N = 1, 2, 4, 8 ranks (the curve-sorted query shard of rank 0, device resident):
what the per-rank kernel does to strong scaling, without needing N GPUs.  PROJECTIONS from one GPU, not measurements on N.
  python tools/shard_rate.py                      uniform 10 M, k = 15, kNN + normals (the bench workload; configs[1]/[3] shape)
  python tools/shard_rate.py clustered 1e7 15     configs[3]'s cloud
  python tools/shard_rate.py uniform 5e7 32 stream   configs[4]: every rank rebuilds the whole index (coarse order), then answers its shard"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("point-cloud-processing_amd")
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
stream = len(sys.argv) > 4 and sys.argv[4] == "stream"
dev = torch.device("cuda", 0)
pts = pkg.synthetic.uniform_cloud(n, 43) if kind == "uniform" else pkg.synthetic.clustered_cloud(n, 44)
d_pts = torch.from_numpy(pts).to(dev)
ix = pkg.Index.from_device(d_pts.data_ptr(), n, device=0, stream=torch.cuda.current_stream().cuda_stream, coarse_order=stream)
d_idx = torch.empty((n, k), dtype=torch.int32, device=dev)
d_cnt = torch.empty(n, dtype=torch.int32, device=dev)
d_nrm = None if stream else torch.empty((n, 3), dtype=torch.float32, device=dev)
res = {"kind": kind, "n": n, "k": k, "step": "rebuild + kNN rows" if stream else "kNN rows + normals", "projection": "one GPU running one rank's share; not a measurement on N GPUs"}


def step(first, count):
    if stream:
        ix.rebuild_dev(d_pts.data_ptr(), n, coarse_order=True)
        ix.knn_self_dev(k, 1e-5, d_idx.data_ptr(), d_cnt.data_ptr(), None, first, count)
    else:
        ix.normals_knn_self_dev(k, 1e-5, d_nrm.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr(), first, count)


for world in (1, 2, 4, 8):
    first, count = pkg.shard_range(n, 0, world)
    for _ in range(3):
        step(first, count)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5 if n > 20_000_000 else 20
    for _ in range(reps):
        step(first, count)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res["ranks_%d" % world] = {"queries": count, "ms": round(dt * 1e3, 4), "mqps_per_rank": round(count / dt / 1e6, 1)}
base = res["ranks_1"]["ms"]
for world in (2, 4, 8):
    res["ranks_%d" % world]["speedup_if_all_ranks_alike"] = round(base / res["ranks_%d" % world]["ms"], 2)
print(json.dumps(res))
