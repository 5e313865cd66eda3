#!/usr/bin/env python3
"""Time of ONE rank's share of the bench step for N = 1, 2, 4, 8 ranks (device resident), every rank of N in turn on this one
GPU: what the per-rank work does to strong scaling, without needing N GPUs.  PROJECTIONS from one GPU, not measurements on N.
The job's step time at N ranks is the SLOWEST rank's; both that and the mean are reported.
  python tools/shard_rate.py                          uniform 10 M, k = 15, kNN + normals (the bench workload; configs[1]/[3] shape)
  python tools/shard_rate.py clustered 1e7 15         configs[3]'s cloud
  python tools/shard_rate.py uniform 5e7 32 stream    configs[4]: every step rebuilds the index, then answers the rank's shard
  ... replicated                                       (last argument) every rank indexes the WHOLE cloud (rounds 1-3) instead of
                                                       building the rank-local index (PCPX_BUILD_SHARD, round 4)"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module("point-cloud-processing_amd")
args = [a for a in sys.argv[1:] if a != "replicated"]
replicated = "replicated" in sys.argv[1:]
kind = args[0] if len(args) > 0 else "uniform"
n = int(float(args[1])) if len(args) > 1 else 10_000_000
k = int(args[2]) if len(args) > 2 else 15
stream = len(args) > 3 and args[3] == "stream"
dev = torch.device("cuda", 0)
pts = pkg.synthetic.uniform_cloud(n, 43) if kind == "uniform" else pkg.synthetic.clustered_cloud(n, 44)
grid = np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32)
d_pts = torch.from_numpy(pts).to(dev)
cs = torch.cuda.current_stream().cuda_stream
kcap = 8 if k <= 8 else 16 if k <= 16 else 32
pitch = kcap if k in (kcap - 1, kcap) else 0  # (rows of 16 entries for k = 15: one aligned 64-byte piece each, bench.py: row_pitch)
d_idx = torch.empty((n, pitch or k), dtype=torch.int32, device=dev)
d_cnt = torch.empty(n, dtype=torch.int32, device=dev)
d_nrm = None if stream else torch.empty((n, 3), dtype=torch.float32, device=dev)
res = {"kind": kind, "n": n, "k": k, "step": "rebuild + kNN rows" if stream else "kNN rows + normals",
       "index": "whole cloud on every rank" if replicated else "rank-local (PCPX_BUILD_SHARD); N = 1: the whole cloud",
       "projection": "one GPU running each rank's share in turn; not a measurement on N GPUs"}
reps = 5 if n > 20_000_000 else 20

by_work = not stream and "count" not in sys.argv[1:]  # static index: shards of equal estimated work (pcpx_shard_cuts_by_cost)
res["cut"] = "equal estimated work" if by_work else "equal query counts"
events = None
if by_work:
    whole = pkg.Index.from_device(d_pts.data_ptr(), n, device=0, stream=cs, voxel_grid=grid)
    events = whole.knn_group_costs(k, 1e-5, 16)
    whole.close()
for world in (1, 2, 4, 8):
    per_rank, trees, build_ms, firsts = [], [], [], []
    cuts = pkg.shard_cuts_by_cost(n, world, 16, events) if by_work else [pkg.shard_range(n, r, world)[0] for r in range(world)] + [n]
    for rank in range(world):
        local = world > 1 and not replicated
        first, count = cuts[rank], cuts[rank + 1] - cuts[rank]
        kw = dict(voxel_grid=grid, shard=(rank, world) if local else None, k_hint=k, borrow=local)
        if local and by_work:
            kw["shard_range"] = (first, count)
        ix = pkg.Index.from_device(d_pts.data_ptr(), n, device=0, stream=cs, **kw)

        def step():
            if stream:
                ix.rebuild_dev(d_pts.data_ptr(), n, **kw)
                ix.knn_self_strided_dev(k, 1e-5, pitch, d_idx.data_ptr(), d_cnt.data_ptr(), None, first, count)
            else:
                ix.normals_knn_self_strided_dev(k, 1e-5, pitch, d_nrm.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr(), first, count)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        firsts.append((time.perf_counter() - t0) * 1e3)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        per_rank.append((time.perf_counter() - t0) / reps * 1e3)
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.rebuild_dev(d_pts.data_ptr(), n, **kw)
        torch.cuda.synchronize()
        build_ms.append((time.perf_counter() - t0) / reps * 1e3)
        if local:
            trees.append(ix.shard_info()["local_points"])
        ix.close()
    r = {"queries_per_rank": n // world, "ms_slowest_rank": round(max(per_rank), 4), "ms_mean_rank": round(sum(per_rank) / world, 4),
         "ms_first_step_slowest_rank (coverage check + recording)": round(max(firsts), 4),
         "index_build_ms_slowest": round(max(build_ms), 4), "index_build_ms_mean": round(sum(build_ms) / world, 4)}
    if trees:
        r["local_tree_points_max"] = max(trees)
        r["local_tree_fraction_of_cloud_max"] = round(max(trees) / n, 4)
    res["ranks_%d" % world] = r
base = res["ranks_1"]["ms_slowest_rank"]
for world in (2, 4, 8):
    res["ranks_%d" % world]["speedup_projected"] = round(base / res["ranks_%d" % world]["ms_slowest_rank"], 2)
    res["ranks_%d" % world]["speedup_projected_mean_rank"] = round(base / res["ranks_%d" % world]["ms_mean_rank"], 2)
print(json.dumps(res))
