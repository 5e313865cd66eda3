#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv) into a short table with readable names."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-46s %6s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for r in rows:
    n = r["Name"]
    n = n.replace("pcpx::(anonymous namespace)::", "")  # (also inside template argument lists)
    m = re.search(r"\b(k_[a-z_0-9]+(<[^>(]*>)?)", n)
    if m: n = m.group(1)
    elif "radix_sort_onesweep_iteration" in n: n = "rocprim::radix_sort_onesweep_iteration"
    elif "onesweep_histograms" in n: n = "rocprim::radix_sort_onesweep_histograms"
    elif "rocprim" in n: n = "rocprim::(other)"
    else: n = n[:46]
    print("%-46s %6s %14s %12.0f %7.2f" % (n[:46], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), float(r["Percentage"])))
