#!/usr/bin/env python3
"""Issue-cycle model of k_knn (DESIGN.md "Roofline"), vector pipe and scalar pipe side by side.
Vector: per-group event counts of the diagnostic build (profiles/<tag>_knn_phase_stats_uniform.json) x the VALU instructions each
event issues (counted in the ISA) x the measured issue cost of each instruction class (profiles/<tag>_valu_issue_rates.txt).
Scalar: the launch's scalar-pipe instruction count from the counters (profiles/<tag>_pmc_summary.json: SALU + SMEM + branches +
whatever SQ_INSTS has beyond the vector / LDS / memory classes: s_waitcnt, s_nop) x the measured 4.1 cycles of an s_add_u32.
Both against the SIMD cycles available per group at the measured kernel duration (profiles/<tag>_bench.json).
Writes profiles/<tag>_valu_issue_model.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))
st = json.load(open(P("knn_phase_stats_uniform.json")))["per_group"]
bench = json.loads(open(P("bench.json")).read().strip().splitlines()[-1])
SIMPLE, THREE_OP, CMP, F64, SCALAR = 2.4, 4.4, 4.1, 4.2, 4.1  # cycles per wave-instruction per SIMD (valu_issue_rates.txt, >= 2 waves resident)
candidate = 10 * SIMPLE + CMP                                 # distance (8) + v_cmpx + address and position bumps (the eps-box test waits for the compaction)
box = 14 * SIMPLE + 3 * THREE_OP + CMP                        # 6 sub, 3 mul, 3 add, poison add, ... + 3 v_max3 + v_cmp (16 VALU)
ce = 2 * F64                                                  # compare-exchange = v_min_f64 + v_max_f64
chunk = (19 + 28) * ce + 8 * F64 + 30   # one chunk of 8 keys: 19-comparator sort, 8 mins, merge of 16 less the 4 sentinel exchanges (k = 15)
second = (1 + 28) * ce + 2 * F64 + 10   # rows 8 .. 9: one compare-exchange, 2 mins, the merge again
short_share = 0.85  # share of compactions in which no lane holds more than 8 keys (the trigger is "more than 2")
sparse, owners = st.get("sparse_leaves", 0.0), st.get("sparse_leaf_lanes", 0.0)  # leaves looked at point-per-lane, and the lanes they were looked at for
leaves = st["leaves"] + st["seed_leaves"] - sparse
owner = 5 * CMP + 8 * SIMPLE + CMP + CMP + THREE_OP + CMP  # 5 v_readlane, the distance, v_cmpx, v_mbcnt, address, v_writelane
parts = {
    "leaf_candidates": leaves * 8 * candidate,
    "sparse_leaf_candidates": sparse * (2 * THREE_OP + 4 * SIMPLE) + owners * owner,
    "box_tests": st["expansions"] * 4 * box,
    "compactions": st["compactions"] * (chunk + (1 - short_share) * second),
    "cap_epilogue_setup": 6000.0,
}
groups = bench["config"]["queries"] / 64
clock_hz, simds = 2.4e9, 1024
available = bench["roofline"]["avg_launch_ms"] * 1e-3 * clock_hz * simds / groups
total = sum(parts.values())
out = {"kernel": "k_knn<16,true,0,false,false,1>", "workload": bench["config"]["workload"],
       "valu_issue_cycles_per_group": {k: round(v) for k, v in parts.items()}, "valu_issue_cycles_per_group_total": round(total),
       "simd_cycles_available_per_group": round(available), "valu_issue_frac": round(total / available, 3),
       "note": "issue costs were measured on streams of one instruction each; a value near 1 says the pipe is saturated, not that the model is exact",
       "assumptions": {"clock_GHz": 2.4, "simds": simds, "short_compaction_share": short_share,
                       "issue_cost_cycles": {"simple_vop2": SIMPLE, "three_operand": THREE_OP, "compare": CMP, "f64_min_max": F64, "scalar": SCALAR}}}
try:
    pmc = json.load(open(P("pmc_summary.json")))
    k = next(v for name, v in pmc.items() if name.startswith("k_knn<16,true,"))
    c = lambda n: k[n]["avg_per_dispatch"] if n in k else 0.0
    other = max(0.0, c("SQ_INSTS") - c("SQ_INSTS_VALU") - c("SQ_INSTS_SALU") - c("SQ_INSTS_SMEM") - c("SQ_INSTS_LDS") - c("SQ_INSTS_BRANCH")) if c("SQ_INSTS") else 0.0
    scalar = c("SQ_INSTS_SALU") + c("SQ_INSTS_SMEM") + c("SQ_INSTS_BRANCH") + other
    out["counters_per_group"] = {"valu": round(c("SQ_INSTS_VALU") / groups), "salu": round(c("SQ_INSTS_SALU") / groups), "smem": round(c("SQ_INSTS_SMEM") / groups),
                                 "lds": round(c("SQ_INSTS_LDS") / groups), "branch": round(c("SQ_INSTS_BRANCH") / groups),
                                 "other (s_waitcnt, s_nop, vector memory ...)": round(other / groups)}
    out["scalar_pipe_cycles_per_group"] = round(scalar / groups * SCALAR)
    out["scalar_pipe_frac"] = round(scalar / groups * SCALAR / available, 3)
except Exception as e:  # counters not collected yet
    out["scalar_pipe_note"] = "no counters: %r" % (e,)
if tag >= "r04":
    # Round 4: the kernel no longer fits a one-pipe model (profiles/experiments/README.md, round 4: 19 % fewer vector instructions bought
    # 3 %, the same count of scalar ones taken away bought 5 %, packed-float instructions nothing).  What is reported from round 4 on is
    # the instruction mix per group from the counters and the cycles a SIMD has per group; the per-class vector model above is kept
    # for rounds 1-3 only (its "point-per-lane leaf" class no longer exists: statistics [14] / [15] count the packed leaves now).
    for k in ("valu_issue_cycles_per_group", "valu_issue_cycles_per_group_total", "valu_issue_frac", "note", "scalar_pipe_cycles_per_group", "scalar_pipe_frac"):
        out.pop(k, None)
    out["assumptions"].pop("short_compaction_share", None)
    if "counters_per_group" in out:
        out["instructions_per_group"] = sum(out["counters_per_group"].values())
        out["simd_cycles_per_instruction_at_7_waves"] = round(available / out["instructions_per_group"], 2)
    out["events_per_group"] = {k: st[k] for k in ("leaves", "expansions", "compactions", "seed_leaves", "seed_compactions", "sparse_leaves", "sparse_leaf_lanes") if k in st}
    out["events_note"] = "diagnostic build (tools/knn_stats.py); sparse_leaves / sparse_leaf_lanes = the leaves looked at in the packed form and the lanes that needed them"
    out["reading"] = ("a wave issues an instruction every ~4 cycles whatever its kind and its stream is largely dependent; with seven waves per SIMD the "
                      "kernel's time follows the instructions per group, the scalar ones first (they wait on each other through SCC and the scalar registers)")
json.dump(out, open(P("valu_issue_model.json"), "w"), indent=1)
print(json.dumps(out))
