#!/usr/bin/env python3
"""Copies the summaries tools/collect_profiles.sh left in gpurun_out/<tag>/ into profiles/ (tracked).
usage: tools/collect_profiles.py r01"""
import glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")

def cp(a, b):
    if os.path.exists(os.path.join(src, a)):  # (the script also runs on the GPU box half way through the collection)
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, "%s_%s" % (tag, b)))

cp("bench.json", "bench.json")
cp("bench_under_rocprofv3.json", "bench_under_rocprofv3.json")
for name, out in (("stats_uniform.json", "knn_phase_stats_uniform.json"), ("stats_clustered.json", "knn_phase_stats_clustered.json"),
                  ("bench_clustered_10m_k15.json", "bench_clustered_10m_k15.json"),
                  ("bench_uniform_10m_k8.json", "bench_uniform_10m_k8.json"),
                  ("bench_c5_50m_k32_stream.json", "bench_c5_50m_k32_stream.json"), ("pcie_inclusive.json", "pcie_inclusive.json"),
                  ("batch_query_rate.json", "batch_query_rate.json"), ("latency.json", "latency.json"),
                  ("shard_rate.json", "shard_rate.json"), ("shard_rate_clustered.json", "shard_rate_clustered_10m_k15.json"),
                  ("shard_rate_c5.json", "shard_rate_c5_50m_k32_stream.json"), ("pcie_rate.json", "pcie_rate.json"),
                  ("rebuild_10m.json", "rebuild_10m.json"), ("rebuild_50m.json", "rebuild_50m.json"),
                  ("rebuild_10m_clustered.json", "rebuild_10m_clustered.json"), ("rebuild_10m_coarse.json", "rebuild_10m_coarse_order.json"),
                  ("rebuild_50m_coarse.json", "rebuild_50m_coarse_order.json"),
                  ("pmc_range/range_kernel_stats.txt", "range_kernel_stats.txt"), ("pmc_range/range_under_prof.json", "range_count_10m.json"),
                  ("pmc_range_pos/range_under_prof.json", "range_count_10m_curve_positions.json"),
                  ("valu_issue_rates.txt", "valu_issue_rates.txt"), ("pmc_latency/latency_kernel_stats.txt", "latency_kernel_stats.txt"),
                  ("pmc_latency/latency_under_prof.json", "latency_under_rocprofv3.json"), ("filter_bench.json", "filter_bench.json"),
                  ("fuzz_filters.json", "fuzz_filters.json"), ("range_lists.json", "range_lists_10m.json"),
                  ("sizes.json", "knn_rate_by_cloud_size.json"), ("outliers_clustered.json", "longest_groups_clustered.json")):
    if os.path.exists(os.path.join(src, name)):
        cp(name, out)
# (bench.py starts one child process for the host-pointer ABI side measurement; rocprofv3 writes a file per process: the
#  main process is the one that finishes last)
stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
if stats:
    shutil.copyfile(stats[0], os.path.join(dst, tag + "_rocprofv3_kernel_stats.csv"))
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), stats[0]], capture_output=True, text=True).stdout
    open(os.path.join(dst, tag + "_rocprofv3_kernel_stats.txt"), "w").write(
        "rocprofv3 --kernel-trace --stats -- python3 bench.py   (MI355X, tools/collect_profiles.sh)\n" + txt)
rstats = sorted(glob.glob(os.path.join(src, "trace_rebuild", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
if rstats:
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), rstats[0]], capture_output=True, text=True).stdout
    open(os.path.join(dst, tag + "_rebuild_kernel_stats.txt"), "w").write(
        "rocprofv3 --kernel-trace --stats -- python3 tools/rebuild_loop.py 1e7 10   (12 rebuilds of 10 M points, auto bounding box)\n" + txt)
r50 = sorted(glob.glob(os.path.join(src, "trace_rebuild50", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
if r50:
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), r50[0]], capture_output=True, text=True).stdout
    open(os.path.join(dst, tag + "_rebuild_50m_kernel_stats.txt"), "w").write(
        "rocprofv3 --kernel-trace --stats -- python3 tools/rebuild_loop.py 5e7 5   (7 rebuilds of 50 M points, auto bounding box)\n" + txt)
for sub, name in (("pmc_rebuild", "pmc_rebuild"), ("pmc_range", "pmc_range"), ("pmc_range_pos", "pmc_range_curve_positions"), ("pmc_latency", "pmc_latency")):
    f = os.path.join(src, sub, "pmc_summary.json")
    if os.path.exists(f):
        d = json.load(open(f))
        keep = {k: {c: round(v["avg_per_dispatch"], 1) for c, v in cs.items()} for k, cs in d.items() if k.startswith("k_")}
        json.dump({"command": "tools/%s.sh: rocprofv3 --kernel-trace --pmc <one counter group per pass> (averages per dispatch; FETCH_SIZE / WRITE_SIZE in KB)" % sub,
                   "fetch_size_rule": "bytes read = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM: gfx950 tallies 128-B requests at 64 B); calibrated here on "
                                      "k_bbox, which reads exactly 12 B per point: 120 MB at 10 M points against FETCH_SIZE = 58.8 MB",
                   "counters": keep}, open(os.path.join(dst, "%s_%s.json" % (tag, name)), "w"), indent=1)
fstats = sorted(glob.glob(os.path.join(src, "trace_filter", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
if fstats:
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), fstats[0]], capture_output=True, text=True).stdout
    open(os.path.join(dst, tag + "_filter_kernel_stats.txt"), "w").write(
        "rocprofv3 --kernel-trace --stats -- python3 tools/filter_bench.py 1e7 1   (10 M uniform points, ranges of radius 0.01 / WLOP h = 0.02, "
        "4 iterations per call, 2 calls)\n" + txt)
pmc = os.path.join(src, "pmc", "pmc_summary.json")
if os.path.exists(pmc):
    shutil.copyfile(pmc, os.path.join(dst, tag + "_pmc_summary.json"))
    allk = json.load(open(pmc))
    d = next((v for k, v in allk.items() if k.startswith("k_knn<16,true,") or k == "k_knn"), {})
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f, w = d["FETCH_SIZE"]["avg_per_dispatch"], d["WRITE_SIZE"]["avg_per_dispatch"]
        json.dump({
            "workload": "uniform_10m_k15",
            "kernel": "k_knn<16,true,0,false,false,1> (fused kNN k=15 + PCA normals, 10 M queries per launch, persistent grid)",
            "command": "tools/pmc_passes.sh: rocprofv3 --kernel-trace --pmc <one counter group per pass> --output-format csv -- "
                       "python3 bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 (FETCH_SIZE in pass 3, WRITE_SIZE in pass 4)",
            "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
            "correction": "one rule for every kernel of this repository: bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM: "
                          "gfx950 tallies 128-B read requests at 64 B), calibrated on k_bbox, which reads exactly 120 MB at 10 M points and "
                          "reports FETCH_SIZE = 58.8 MB (profiles/%s_pmc_rebuild.json); this kernel's reads are 32-128 B scalar loads and "
                          "4-B gathers, for which the factor is not separately calibrated: an upper estimate" % tag,
            "k_knn_hbm_bytes_per_launch": int((2 * f + w) * 1024),
            "k_knn_hbm_bytes_per_launch_uncorrected": int((f + w) * 1024),
            "algorithmic_bytes_per_launch": 84 * 10_000_000}, open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1)
if os.path.exists(pmc):
    # what bench.py's roofline.issue is made from: instruction counts per launch by issue class, and the shader clock under this load
    # (GRBM_GUI_ACTIVE counts every XCD's busy cycles: / 8; the launch's duration from the trace of the same command)
    allk = json.load(open(pmc))
    d = next((v for k, v in allk.items() if k.startswith("k_knn<16,true,")), {})
    c = lambda name: d[name]["avg_per_dispatch"] if name in d else 0.0
    if c("SQ_INSTS"):
        other = max(0.0, c("SQ_INSTS") - c("SQ_INSTS_VALU") - c("SQ_INSTS_SALU") - c("SQ_INSTS_SMEM") - c("SQ_INSTS_LDS") - c("SQ_INSTS_BRANCH"))
        clock = 2.4e9
        stats_csv = os.path.join(dst, tag + "_rocprofv3_kernel_stats.csv")
        if c("GRBM_GUI_ACTIVE") and os.path.exists(stats_csv):
            import csv
            for row in csv.DictReader(open(stats_csv)):
                if "k_knn<16, true," in row["Name"] or "k_knn<16,true," in row["Name"].replace(" ", ""):
                    clock = c("GRBM_GUI_ACTIVE") / 8.0 / (float(row["AverageNs"]) * 1e-9)
                    break
        json.dump({"workload": "uniform_10m_k15", "kernel": "k_knn<16,true,0,false,false,1>", "simds": 1024,
                   "instructions_per_launch": c("SQ_INSTS"), "valu_instructions_per_launch": c("SQ_INSTS_VALU"),
                   "scalar_class_instructions_per_launch": c("SQ_INSTS_SALU") + c("SQ_INSTS_SMEM") + c("SQ_INSTS_BRANCH") + other,
                   "scalar_class": {"salu": c("SQ_INSTS_SALU"), "smem": c("SQ_INSTS_SMEM"), "branch": c("SQ_INSTS_BRANCH"),
                                    "other (s_waitcnt, s_nop, s_setprio, vector memory)": other},
                   "lds_instructions_per_launch": c("SQ_INSTS_LDS"),
                   "cycles_per_scalar_instruction": 4.1, "cycles_per_valu_instruction_range": [2.4, 4.3],
                   "issue_cost_source": "profiles/r03_valu_issue_rates.txt (tools/valu_rate.hip: s_add_u32 4.04-4.21 cycles per SIMD at 1-5 waves; v_add / v_mul / "
                                        "v_or 2.2-2.7 at >= 2 waves, three-operand / compare / f64 / packed 4.1-4.5)",
                   "shader_clock_hz": clock, "shader_clock_source": "GRBM_GUI_ACTIVE / 8 per launch over the launch's average duration in "
                                                                     "profiles/%s_rocprofv3_kernel_stats.csv (2.4e9 if either is missing)" % tag,
                   "command": "tools/pmc_passes.sh (rocprofv3 --kernel-trace --pmc, one counter group per pass) -- python3 bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1"},
                  open(os.path.join(dst, tag + "_issue.json"), "w"), indent=1)
pos = os.path.join(src, "pmc_pos", "pmc_summary.json")
if os.path.exists(pos):
    allk = json.load(open(pos))
    d = next((v for k, v in allk.items() if k.startswith("k_knn<16,true,") or k == "k_knn"), {})
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f, w = d["FETCH_SIZE"]["avg_per_dispatch"], d["WRITE_SIZE"]["avg_per_dispatch"]
        json.dump({"workload": "uniform_10m_k15, rows and normals at curve positions (pcpx_knn_self_curve_order_dev)",
                   "command": "tools/pmc_passes.sh ... -- python3 bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 --rows-at-curve-positions",
                   "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "WRITE_bytes_per_launch": int(w * 1024),
                   "payload_bytes_per_launch": 76 * 10_000_000,
                   "counters": {c: round(v["avg_per_dispatch"], 1) for c, v in d.items()}}, open(os.path.join(dst, tag + "_pmc_rows_at_curve_positions.json"), "w"), indent=1)
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_issue_model.py"), tag], check=False)
print("copied into", dst)
