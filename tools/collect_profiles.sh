#!/bin/bash
# Collects every measurement DESIGN.md quotes, on the GPU box, into gpurun_out/<tag>/ :
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r05 [part]'      part: all (default) | knn (= knn1 + knn2) | knn1 | knn2 | build | host
#   (all of it does not fit one 20-minute call: knn1, knn2, host, build are four)
# then, back in the container:  python3 tools/collect_profiles.py r05   (copies the summaries into profiles/)
set -u
tag=${1:-r05}
part=${2:-all}
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$part" = all ] || [ "$part" = knn ] || [ "$part" = knn1 ]; then
# 1. PMC passes of the bench command (separate runs per counter group, --kernel-trace only)
bash tools/pmc_passes.sh "$out/pmc" --steps 3 --warmup 1 > "$out/pmc.txt" 2>&1 || { echo "pmc failed"; tail -5 "$out/pmc.txt"; }
echo "pmc done"
# 1b. the same launch with rows and normals written at curve positions (pcpx_knn_self_curve_order_dev): its memory-side bytes
PMC_PASSES=5 bash tools/pmc_passes.sh "$out/pmc_pos" --steps 3 --warmup 1 --rows-at-curve-positions > "$out/pmc_pos.txt" 2>&1 || { echo "pmc (curve positions) failed"; tail -5 "$out/pmc_pos.txt"; }
# (the HBM traffic file bench.py quotes under roofline.traffic is made from these passes before the bench line is taken)
python3 tools/collect_profiles.py "$tag" > /dev/null 2>&1
# 2. the default command (python3 bench.py) under rocprofv3 --kernel-trace --stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py > "$out/bench_under_rocprofv3.json" 2> "$out/trace.err" || { echo "trace failed"; tail -5 "$out/trace.err"; }
echo "trace done"
python3 tools/collect_profiles.py "$tag" > /dev/null 2>&1  # (the shader clock of the issue figure comes from this trace)
# 3. the bench line (default workload, CPU baseline included; --with-1m adds the configs[1] side figure)
timeout -k 10 400 python3 bench.py --with-1m > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; }
echo "bench done"
# 4. traversal statistics + per-phase clocks (diagnostic build of the kernel)
timeout -k 10 200 python3 tools/knn_stats.py 1e7 uniform 15 > "$out/stats_uniform.json" 2>> "$out/stats.err" &&
timeout -k 10 200 python3 tools/knn_stats.py 1e7 clustered 15 > "$out/stats_clustered.json" 2>> "$out/stats.err" || echo "stats failed"
echo "stats done"
fi
if [ "$part" = all ] || [ "$part" = knn ] || [ "$part" = knn2 ]; then
# 5. other workloads: configs[3]'s cloud on one GPU, configs[4] streaming
timeout -k 10 300 python3 bench.py --workload clustered_10m_k15 --no-cpu-baseline > "$out/bench_clustered_10m_k15.json" 2>> "$out/bench.err" || echo "clustered failed"
timeout -k 10 300 python3 bench.py --workload uniform_10m_k8 --no-cpu-baseline --no-extra > "$out/bench_uniform_10m_k8.json" 2>> "$out/bench.err" || echo "k8 failed"
timeout -k 10 500 python3 bench.py --workload uniform_50m_k32_stream --no-cpu-baseline --steps 5 > "$out/bench_c5_50m_k32_stream.json" 2>> "$out/bench.err" || echo "c5 failed"
echo "workloads done"
timeout -k 10 300 python3 tools/shard_rate.py > "$out/shard_rate.json" 2>> "$out/bench.err" || echo "shard failed"
timeout -k 10 300 python3 tools/shard_rate.py clustered 1e7 15 > "$out/shard_rate_clustered.json" 2>> "$out/bench.err" || echo "shard clustered failed"
timeout -k 10 400 python3 tools/shard_rate.py uniform 5e7 32 stream > "$out/shard_rate_c5.json" 2>> "$out/bench.err" || echo "shard c5 failed"
timeout -k 10 300 python3 tools/batch_query_rate.py > "$out/batch_query_rate.json" 2>> "$out/bench.err" || echo "batch failed"
# round 5: lists of every point's range (counts + scan + fill) and 1 M boxes; the step at several cloud sizes; the longest query groups
timeout -k 10 300 python3 tools/range_lists_rate.py > "$out/range_lists.json" 2>> "$out/bench.err" || echo "range lists failed"
timeout -k 10 300 python3 tools/ab_sizes.py 15 > "$out/sizes.json" 2>> "$out/bench.err" || echo "sizes failed"
timeout -k 10 300 python3 tools/outliers.py clustered 1e7 15 > "$out/outliers_clustered.json" 2>> "$out/bench.err" || echo "outliers failed"
# configs[2]'s kernel: per-kernel time and counters
bash tools/pmc_range.sh "$out/pmc_range" > "$out/pmc_range.txt" 2>&1 || echo "pmc range failed"
RANGE_FORM=curve bash tools/pmc_range.sh "$out/pmc_range_pos" > "$out/pmc_range_pos.txt" 2>&1 || echo "pmc range (curve positions) failed"
echo "range done"
fi
if [ "$part" = all ] || [ "$part" = host ]; then
# 6. host-pointer ABI (PCIe inclusive), single-query / range / construction latency, PCIe line rate
timeout -k 10 300 python3 tools/pcie_inclusive.py > "$out/pcie_inclusive.json" 2>> "$out/bench.err" || echo "pcie failed"
timeout -k 10 600 python3 tools/latency_report.py "$out/latency.json" > "$out/latency.log" 2>&1 || { echo "latency failed"; tail -3 "$out/latency.log"; }
hipcc -O2 --offload-arch=gfx950 tools/pcie_rate.hip -o /tmp/pcie_rate -pthread && timeout -k 10 120 /tmp/pcie_rate > "$out/pcie_rate.json" 2>> "$out/bench.err"
bash tools/pmc_latency.sh "$out/pmc_latency" > "$out/pmc_latency.txt" 2>&1 || echo "pmc latency failed"
echo "host side done"
fi
if [ "$part" = all ] || [ "$part" = build ]; then
# 7. the rebuild: times (default and PCPX_BUILD_COARSE_ORDER), per-kernel times at 10 M and 50 M, counters at 10 M
timeout -k 10 300 python3 tools/rebuild_loop.py 1e7 20 > "$out/rebuild_10m.json" 2>> "$out/bench.err"
timeout -k 10 300 python3 tools/rebuild_loop.py 1e7 20 clustered > "$out/rebuild_10m_clustered.json" 2>> "$out/bench.err"
timeout -k 10 300 python3 tools/rebuild_loop.py 5e7 10 > "$out/rebuild_50m.json" 2>> "$out/bench.err"
timeout -k 10 300 python3 tools/rebuild_loop.py 1e7 20 uniform coarse > "$out/rebuild_10m_coarse.json" 2>> "$out/bench.err"
timeout -k 10 300 python3 tools/rebuild_loop.py 5e7 10 uniform coarse > "$out/rebuild_50m_coarse.json" 2>> "$out/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_rebuild" -- python3 tools/rebuild_loop.py 1e7 10 > "$out/rebuild_under_prof.json" 2>> "$out/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_rebuild50" -- python3 tools/rebuild_loop.py 5e7 5 > "$out/rebuild50_under_prof.json" 2>> "$out/bench.err"
bash tools/pmc_rebuild.sh "$out/pmc_rebuild" > "$out/pmc_rebuild.txt" 2>&1 || echo "pmc rebuild failed"
echo "rebuild done"
# 8. instruction issue costs
hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && timeout -k 10 120 /tmp/valu_rate > "$out/valu_issue_rates.txt" 2>&1
fi
echo "all done"
