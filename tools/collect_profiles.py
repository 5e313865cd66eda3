#!/usr/bin/env python3
"""Copies the summaries tools/collect_profiles.sh left in gpurun_out/<tag>/ into profiles/ (tracked).
usage: tools/collect_profiles.py r01"""
import glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
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
                  ("shard_rate.json", "shard_rate.json"), ("pcie_rate.json", "pcie_rate.json"),
                  ("rebuild_10m.json", "rebuild_10m.json"), ("rebuild_50m.json", "rebuild_50m.json"),
                  ("valu_issue_rates.txt", "valu_issue_rates.txt"), ("filter_bench.json", "filter_bench.json"),
                  ("fuzz_filters.json", "fuzz_filters.json")):
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
fstats = sorted(glob.glob(os.path.join(src, "trace_filter", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
if fstats:
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_summary.py"), fstats[0]], capture_output=True, text=True).stdout
    open(os.path.join(dst, tag + "_filter_kernel_stats.txt"), "w").write(
        "rocprofv3 --kernel-trace --stats -- python3 tools/filter_bench.py 1e7 1   (10 M uniform points, ranges of radius 0.01 / WLOP h = 0.02, "
        "4 iterations per call, 2 calls)\n" + txt)
pmc = os.path.join(src, "pmc", "pmc_summary.json")
if os.path.exists(pmc):
    shutil.copyfile(pmc, os.path.join(dst, tag + "_pmc_summary.json"))
    d = json.load(open(pmc)).get("k_knn", {})
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f, w = d["FETCH_SIZE"]["avg_per_dispatch"], d["WRITE_SIZE"]["avg_per_dispatch"]
        json.dump({
            "workload": "uniform_10m_k15",
            "kernel": "k_knn<16,true,false,false> (fused kNN k=15 + PCA normals, 10 M queries per launch, persistent grid)",
            "command": "tools/pmc_passes.sh: rocprofv3 --kernel-trace --pmc <one counter group per pass> --output-format csv -- "
                       "python3 bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 (FETCH_SIZE in pass 3, WRITE_SIZE in pass 4)",
            "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
            "correction": "MI355X_MICROARCH.md section HBM: bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, and on gfx950 FETCH_SIZE reports "
                          "1/2 of the bytes of a wide coalesced read, so the read side is doubled; this kernel's reads are 32-128 B "
                          "scalar loads and 4-B gathers, an access shape the guide marks uncalibrated, so the doubled figure is an "
                          "upper estimate",
            "k_knn_hbm_bytes_per_launch": int((2 * f + w) * 1024),
            "k_knn_hbm_bytes_per_launch_uncorrected": int((f + w) * 1024),
            "algorithmic_bytes_per_launch": 84 * 10_000_000}, open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1)
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_issue_model.py"), tag], check=False)
print("copied into", dst)
