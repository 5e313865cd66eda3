#!/bin/bash
# Collects every measurement DESIGN.md quotes, on the GPU box, into gpurun_out/<tag>/ :
#   gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh r02'
# then, back in the container:  python3 tools/collect_profiles.py r02   (copies the summaries into profiles/)
set -u
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# 1. PMC passes (separate runs per counter group, --kernel-trace only)
bash tools/pmc_passes.sh "$out/pmc" --steps 3 --warmup 1 > "$out/pmc.txt" 2>&1 || { echo "pmc failed"; tail -5 "$out/pmc.txt"; exit 1; }
echo "pmc done"
# (the HBM traffic file bench.py quotes under roofline.traffic is made from these passes before the bench line is taken)
python3 tools/collect_profiles.py "$tag" > /dev/null 2>&1
# 2. the bench line (default workload, CPU baseline included; --with-1m adds the configs[1] side figure)
timeout -k 10 400 python3 bench.py --with-1m > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
echo "bench done"
# 3. the default command (python3 bench.py) under rocprofv3 --kernel-trace --stats
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py > "$out/bench_under_rocprofv3.json" 2> "$out/trace.err" || { echo "trace failed"; tail -5 "$out/trace.err"; exit 1; }
echo "trace done"
# 4. traversal statistics + per-phase clocks (diagnostic build of the kernel)
timeout -k 10 200 python3 tools/knn_stats.py 1e7 uniform 15 > "$out/stats_uniform.json" 2>> "$out/stats.err" &&
timeout -k 10 200 python3 tools/knn_stats.py 1e7 clustered 15 > "$out/stats_clustered.json" 2>> "$out/stats.err" || { echo "stats failed"; exit 1; }
echo "stats done"
# 5. other workloads: configs[3]'s cloud on one GPU, configs[4] streaming
timeout -k 10 300 python3 bench.py --workload clustered_10m_k15 --no-cpu-baseline > "$out/bench_clustered_10m_k15.json" 2>> "$out/bench.err" &&
timeout -k 10 300 python3 bench.py --workload uniform_10m_k8 --no-cpu-baseline --no-extra > "$out/bench_uniform_10m_k8.json" 2>> "$out/bench.err" &&
timeout -k 10 500 python3 bench.py --workload uniform_50m_k32_stream --no-cpu-baseline --steps 5 > "$out/bench_c5_50m_k32_stream.json" 2>> "$out/bench.err" || { echo "workloads failed"; exit 1; }
echo "workloads done"
# 6. host-pointer ABI (PCIe inclusive), arbitrary query batches, single-query latency, per-rank shard time, PCIe line rate
timeout -k 10 300 python3 tools/pcie_inclusive.py > "$out/pcie_inclusive.json" 2>> "$out/bench.err" || { echo "pcie failed"; exit 1; }
timeout -k 10 300 python3 tools/batch_query_rate.py > "$out/batch_query_rate.json" 2>> "$out/bench.err" || { echo "batch failed"; exit 1; }
timeout -k 10 600 python3 tools/latency_report.py "$out/latency.json" > "$out/latency.log" 2>&1 || { echo "latency failed"; tail -3 "$out/latency.log"; }
timeout -k 10 300 python3 tools/shard_rate.py > "$out/shard_rate.json" 2>> "$out/bench.err" || echo "shard failed"
hipcc -O2 --offload-arch=gfx950 tools/pcie_rate.hip -o /tmp/pcie_rate -pthread && timeout -k 10 120 /tmp/pcie_rate > "$out/pcie_rate.json" 2>> "$out/bench.err"
echo "host side done"
# 7. the rebuild: per-kernel times (10 M) and the 50 M figure
timeout -k 10 300 python3 tools/rebuild_loop.py 1e7 10 > "$out/rebuild_10m.json" 2>> "$out/bench.err"
timeout -k 10 300 python3 tools/rebuild_loop.py 5e7 5 > "$out/rebuild_50m.json" 2>> "$out/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_rebuild" -- python3 tools/rebuild_loop.py 1e7 10 > "$out/rebuild_under_prof.json" 2>> "$out/bench.err"
echo "rebuild done"
# 8. instruction issue costs
hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && timeout -k 10 120 /tmp/valu_rate > "$out/valu_issue_rates.txt" 2>&1
# 9. the range consumers (bilateral filter, WLOP): throughput and per-kernel times; randomised parity
timeout -k 10 400 python3 tools/filter_bench.py 1e7 3 > "$out/filter_bench.json" 2>> "$out/bench.err" || echo "filter bench failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_filter" -- python3 tools/filter_bench.py 1e7 1 > "$out/filter_under_prof.json" 2>> "$out/bench.err"
timeout -k 10 300 python3 tests/fuzz_filters.py 90 4242 > "$out/fuzz_filters.log" 2>&1; tail -1 "$out/fuzz_filters.log" > "$out/fuzz_filters.json"
echo "all done"
