#!/usr/bin/env python3
"""Actual residency of the persistent k_knn waves, from per-wave start/end timestamps (diagnostic build)."""
import importlib, sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("point-cloud-processing_amd")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
pts = pkg.synthetic.uniform_cloud(n, 43) if kind == "uniform" else pkg.synthetic.clustered_cloud(n, 44)
ix = pkg.Index(pts)
ix.debug_knn_stats(15)
d = ix.debug_knn_stats(15, want_waves=True)
w = d["wave_times"].astype(np.int64)
t0, t1 = w[:, 0].min(), w[:, 1].max()
dur = t1 - t0
ev = np.concatenate([np.stack([w[:, 0], np.ones(len(w), np.int64)], 1), np.stack([w[:, 1], -np.ones(len(w), np.int64)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
conc = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0], append=ev[-1, 0])
mean_conc = float((conc * dt).sum() / max(1, dur))
worked = w[:, 2] > 0
st = np.sort(w[:, 0] - t0) / 1e5
print("start time percentiles (ms):", {q: round(float(np.percentile(st, q)), 3) for q in (1, 10, 25, 50, 75, 90, 99)})
print("started within 0.05/0.2/1/5 ms:", int((st < 0.05).sum()), int((st < 0.2).sum()), int((st < 1).sum()), int((st < 5).sum()))
en = np.sort(w[:, 1] - t0) / 1e5
print("end time percentiles (ms):", {q: round(float(np.percentile(en, q)), 3) for q in (1, 10, 50, 90, 99)})
blk = np.arange(len(w))
order = np.argsort(w[:, 0])
print("first 16 blocks to start:", order[:16].tolist(), " last 8:", order[-8:].tolist())
late = w[w[:, 1] - t0 > 1.05 * np.percentile(w[:, 1] - t0, 90)]
print("waves ending late:", len(late))
top = w[np.argsort(-w[:, 3])][:12]
print("slowest groups (ms, group id, of", int(w[:, 2].sum()), "groups; mean group ms =", round(float((w[:, 1] - w[:, 0])[worked].sum() / max(1, w[:, 2].sum())) / 1e5, 4), "):")
for r in top: print("   ", round(r[3] / 1e5, 3), int(r[4]), " wave end at", round((r[1] - t0) / 1e5, 3))
print("second-round groups:", d.get("second_round_groups"), " leaves/group:", round(d["leaves"] / d["waves"], 1), " compactions/group:", round(d["compactions"] / d["waves"], 1))
print(json.dumps({"waves_launched": int(len(w)), "waves_that_got_work": int(worked.sum()), "kernel_ms": dur / 1e5,
                  "max_concurrent": int(conc.max()), "mean_concurrent": round(mean_conc, 1), "per_CU_mean": round(mean_conc / 256, 2),
                  "start_spread_ms": round(float(np.percentile(w[:, 0] - t0, 99)) / 1e5, 3),
                  "groups_per_working_wave_mean": round(float(w[worked, 2].mean()), 1)}))
