#!/usr/bin/env python3
"""A/B of round 5's output forms and schedules on one box (device resident, HIP events on the launch stream):
   python tools/ab_round5.py [uniform|clustered] [n] [k]
Prints one JSON object.  Also dumps, for the calibration of pcpx_shard_cuts_by_cost's weights, the per-group event counts
(stride 1) beside the per-group recorded times."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module("point-cloud-processing_amd")
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
dev = torch.device("cuda", 0)
pts = pkg.synthetic.uniform_cloud(n, 43) if kind == "uniform" else pkg.synthetic.clustered_cloud(n, 44)
grid = np.concatenate([pts.min(0), pts.max(0)]).astype(np.float32)
d_pts = torch.from_numpy(pts).to(dev)
cs = torch.cuda.current_stream().cuda_stream
kcap = 8 if k <= 8 else 16 if k <= 16 else 32
d_idx = torch.empty((n, kcap), dtype=torch.int32, device=dev)
d_cnt = torch.empty(n, dtype=torch.int32, device=dev)
d_nrm = torch.empty((n, 3), dtype=torch.float32, device=dev)


def time_ms(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 4)


res = {"kind": kind, "n": n, "k": k}
ix = pkg.Index.from_device(d_pts.data_ptr(), n, device=0, stream=cs, voxel_grid=grid)
size = ix.size()
for lpt in (0, 1):
    for gather in (0, 1):
        for stride in (0, kcap):
            ix.debug_set("long_groups_first", lpt)
            ix.debug_set("gather_outputs", gather)
            f = lambda: ix.normals_knn_self_strided_dev(k, 1e-5, stride, d_nrm.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr())
            res["whole lpt=%d gather=%d stride=%d" % (lpt, gather, stride)] = [time_ms(f), time_ms(f)]
f = lambda: ix.knn_self_curve_order_dev(k, 1e-5, d_idx.data_ptr(), d_cnt.data_ptr(), None, d_nrm.data_ptr())
for lpt in (0, 1):
    ix.debug_set("long_groups_first", lpt)
    res["whole curve-order lpt=%d" % lpt] = [time_ms(f), time_ms(f)]
# rows only (no normals)
for stride in (0, kcap):
    ix.debug_set("long_groups_first", 1)
    f = lambda: ix.knn_self_strided_dev(k, 1e-5, stride, d_idx.data_ptr(), d_cnt.data_ptr())
    res["rows only stride=%d" % stride] = time_ms(f)
# the cost sample and what it costs
t0 = time_ms(lambda: ix.knn_group_costs(k, 1e-5, 16), reps=3, warm=1)
res["cost_sample_stride16_ms_incl_download"] = t0
ev = ix.knn_group_costs(k, 1e-5, 16)
cuts = pkg.shard_cuts_by_cost(size, 8, 16, ev)
res["cuts_by_work"] = cuts
# calibration dump: events of every group + recorded times
ix.debug_set("long_groups_first", 1)
ix.debug_set("gather_outputs", 1)
ix.rebuild_dev(d_pts.data_ptr(), n, voxel_grid=grid)
ix.normals_knn_self_strided_dev(k, 1e-5, kcap, d_nrm.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr())
ix.synchronize()
gt = ix.debug_group_times()
ev1 = ix.knn_group_costs(k, 1e-5, 1)
m = min(len(gt), len(ev1))
if m:
    A = np.stack([np.ones(m), ev1[:m, 0], ev1[:m, 1], ev1[:m, 2], ev1[:m, 3] & 0xFFFF, ev1[:m, 3] >> 16], 1).astype(np.float64)
    y = gt[:m].astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, y, rcond=None)
    pred = A @ coef
    res["calibration"] = {"groups": int(m), "mean_ticks64": float(y.mean()), "p50": float(np.median(y)), "p99": float(np.percentile(y, 99)), "max": float(y.max()),
                          "lstsq [const, expansions, dense leaves, packed leaves, packed steps, folds] (ticks/64)": [round(float(c), 3) for c in coef],
                          "r2": round(float(1 - ((y - pred) ** 2).sum() / ((y - y.mean()) ** 2).sum()), 4),
                          "mean events [expansions, dense, packed, steps, folds]": [round(float(v), 2) for v in A[:, 1:].mean(0)]}
    w = np.array([6000, 112, 108, 38, 11, 140], np.float64)
    model = A @ w
    res["calibration"]["corr(model cost, ticks)"] = round(float(np.corrcoef(model, y)[0, 1]), 4)
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez_compressed("gpurun_out/group_events_%s_%d_k%d.npz" % (kind, n, k), events=ev1[:m], ticks=gt[:m])
ix.close()

# one eighth: by count vs by work, lpt off / on (every rank in turn)
for cut_name, bounds in (("count", [pkg.shard_range(size, r, 8)[0] for r in range(8)] + [size]), ("work", cuts)):
    for lpt in (0, 1):
        per_rank, first_call = [], []
        for rank in range(8):
            first, count = bounds[rank], bounds[rank + 1] - bounds[rank]
            sh = pkg.Index.from_device(d_pts.data_ptr(), n, device=0, stream=cs, voxel_grid=grid, shard=(rank, 8), shard_range=(first, count), k_hint=k, borrow=True)
            sh.debug_set("long_groups_first", lpt)
            f = lambda: sh.normals_knn_self_strided_dev(k, 1e-5, kcap, d_nrm.data_ptr(), d_idx.data_ptr(), d_cnt.data_ptr(), first, count)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); torch.cuda.synchronize()
            first_call.append(a.elapsed_time(b))
            per_rank.append(time_ms(f))
            sh.close()
        res["eighth cut=%s lpt=%d" % (cut_name, lpt)] = {"ms_per_rank": per_rank, "slowest": max(per_rank), "mean": round(sum(per_rank) / 8, 4),
                                                         "slowest_over_mean": round(max(per_rank) / (sum(per_rank) / 8), 3),
                                                         "first_call_ms_slowest (with the coverage check)": round(max(first_call), 3)}
print(json.dumps(res))
