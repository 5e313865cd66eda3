#!/usr/bin/env python3
"""Extracts one kernel's ISA from `hipcc -S --cuda-device-only` output and prints instruction-class counts
per basic block (VALU / SALU / SMEM / LDS / VMEM / scratch), to see where a hot loop spends its issue slots.
usage: tools/isa_extract.py file.s <substring of the mangled kernel name> [out.s]"""
import re
import sys

def classify(op):
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    return None

def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if pat in l and re.match(r"^[_A-Za-z0-9]+:", l))
    end = next(i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end"))
    body = lines[start:end]
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("\n".join(body))
    block, counts, order = "entry", {}, []
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            block = m.group(1)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")): continue
        c = classify(t.split()[0])
        if c is None: continue
        if block not in counts:
            counts[block] = {}
            order.append(block)
        counts[block][c] = counts[block].get(c, 0) + 1
    tot = {}
    for b in order:
        print(b.ljust(12), " ".join("%s=%d" % kv for kv in sorted(counts[b].items())))
        for k, v in counts[b].items(): tot[k] = tot.get(k, 0) + v
    print("total".ljust(12), " ".join("%s=%d" % kv for kv in sorted(tot.items())))

if __name__ == "__main__":
    main()
