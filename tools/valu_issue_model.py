#!/usr/bin/env python3
"""VALU issue-cycle model of k_knn (DESIGN.md "Roofline"): per-group event counts of the diagnostic build
(profiles/<tag>_knn_phase_stats_uniform.json) x the VALU instructions each event issues (counted in the ISA, tools/isa_extract.py)
x the measured issue cost of each instruction class (profiles/<tag>_valu_issue_rates.txt), against the SIMD cycles available per
group at the measured kernel duration (profiles/<tag>_bench.json).  Writes profiles/<tag>_valu_issue_model.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))
st = json.load(open(P("knn_phase_stats_uniform.json")))["per_group"]
bench = json.load(open(P("bench.json")))
SIMPLE, THREE_OP, CMP, F64 = 2.4, 4.4, 4.1, 4.2  # cycles per wave-instruction per SIMD (valu_issue_rates.txt, >= 2 waves resident)
candidate = 8 * SIMPLE + THREE_OP + 2 * CMP + 2 * SIMPLE      # distance (8) + v_max3 + 2 v_cmpx + address and position bumps
box = 14 * SIMPLE + 3 * THREE_OP + CMP                        # 6 sub, 3 mul, 3 add, poison add, ... + 3 v_max3 + v_cmp (16 VALU)
ce = 2 * F64                                                  # compare-exchange = v_min_f64 + v_max_f64
chunk = (24 + 32) * ce + 8 * F64 + 30   # one chunk of 8 keys: sort 8, 8 mins, merge 16, overhead (empty slots hold PAD_KEY: no masks)
compact_short = chunk          # no lane holds more than 8 keys
compact_full = 2 * chunk       # rows 8..BUF-1 as a second chunk
leaves = st["leaves"] + st["seed_leaves"]
steps = leaves + st["expansions"]
short_share = 0.85  # share of compactions in which no lane holds more than 8 keys (the trigger is "more than 2")
parts = {
    "leaf_candidates": leaves * 8 * candidate,
    "box_tests": st["expansions"] * 4 * box,
    "compactions": st["compactions"] * (short_share * compact_short + (1 - short_share) * compact_full),
    "loop_control": steps * 25.0,
    "cap_epilogue_setup": 4000.0,
}
groups = bench["config"]["queries"] / 64
clock_hz, simds = 2.4e9, 1024
available = bench["roofline"]["avg_launch_ms"] * 1e-3 * clock_hz * simds / groups
total = sum(parts.values())
out = {"kernel": "k_knn<16,true,false,false>", "workload": bench["config"]["workload"],
       "valu_issue_cycles_per_group": {k: round(v) for k, v in parts.items()}, "valu_issue_cycles_per_group_total": round(total),
       "simd_cycles_available_per_group": round(available), "valu_issue_frac": round(total / available, 3),
       "note": "issue costs were measured on streams of one instruction each; a value near 1 says the SIMDs are saturated with vector issue, not that the model is exact",
       "assumptions": {"clock_GHz": 2.4, "simds": simds, "short_compaction_share": short_share,
                       "issue_cost_cycles": {"simple_vop2": SIMPLE, "three_operand": THREE_OP, "compare": CMP, "f64_min_max": F64}}}
json.dump(out, open(P("valu_issue_model.json"), "w"), indent=1)
print(json.dumps(out))
