#!/usr/bin/env python3
"""Does the box of a query group's 64 points predict that the group is a slow one?  (A launch's tail is its slow groups; if
they can be told from their boxes, they can be started first.)  Slow groups: the slowest group of every recorded wave of the
diagnostic kernel (tools/group_time_spread.py).  Prints where those groups rank among all groups by box diagonal.
usage: python tools/group_cost_proxy.py [n] [uniform|clustered] [k]"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
pkg = importlib.import_module("point-cloud-processing_amd")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "clustered"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
pts = pkg.synthetic.uniform_cloud(n, 43) if kind == "uniform" else pkg.synthetic.clustered_cloud(n, 44)
d = torch.from_numpy(pts).cuda()
ix = pkg.Index.from_device(d.data_ptr(), n)
perm = torch.empty(n, dtype=torch.int32, device="cuda")
ix.perm_dev(perm.data_ptr())
ix.synchronize()
st = ix.debug_knn_stats(k, want_waves=True)
w = st["wave_times"].astype(np.float64)
groups = (n + 63) // 64
sp = d[perm.long()]
pad = groups * 64 - n
if pad:
    sp = torch.cat([sp, sp[-1:].expand(pad, 3)])
g = sp.view(groups, 64, 3)
ext = g.max(1).values - g.min(1).values
diag2 = (ext * ext).sum(1).cpu().numpy()
order = np.argsort(diag2)
rank = np.empty(groups, np.int64)
rank[order] = np.arange(groups)
slow = w[:, 4].astype(np.int64)
slow_t = w[:, 3] * 1e-2
mean_group = ((w[:, 1] - w[:, 0]) * 1e-2).sum() / w[:, 2].sum()
pct = rank[slow] / groups
very = slow_t > 3 * mean_group
print(json.dumps({"n": n, "kind": kind, "groups": int(groups), "mean_group_us": round(float(mean_group), 1),
                  "slowest_groups_recorded": int(len(slow)), "of_them_over_3x_mean": int(very.sum()),
                  "box_diagonal_percentile_of_the_over_3x_groups": {p: round(float(np.percentile(pct[very], p)), 4) for p in (5, 25, 50, 75)} if very.any() else None,
                  "box_diagonal_percentile_of_all_slowest_groups": {p: round(float(np.percentile(pct, p)), 4) for p in (5, 25, 50, 75)},
                  "share_of_over_3x_groups_in_top_5pct_by_box": round(float((pct[very] > 0.95).mean()), 3) if very.any() else None,
                  "share_of_over_3x_groups_in_top_20pct_by_box": round(float((pct[very] > 0.80).mean()), 3) if very.any() else None}))
