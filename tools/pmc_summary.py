#!/usr/bin/env python3
"""Condenses the per-dispatch counter CSVs of tools/pmc_passes.sh into per-kernel averages (JSON)."""
import csv, glob, json, os, re, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?|radix_sort_onesweep_iteration|onesweep_histograms)", name)
        key = m.group(1).replace(" ", "") if m else name[:40]
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, d in acc.items():
    res[k] = {c: {"avg_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in d.items()}
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for k in sorted(res):
    if k.startswith("k_knn") or k.startswith("k_normals"):
        print(k, json.dumps({c: round(v["avg_per_dispatch"], 1) for c, v in res[k].items()}))
