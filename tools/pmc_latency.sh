#!/bin/bash
# Kernel times and counters of the latency kernels (k_knn_few, k_range_one): bash tools/pmc_latency.sh <outdir>
set -u
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 tools/latency_loop.py > "$out/latency_under_prof.json" 2> "$out/trace.err" || { echo "trace failed"; tail -3 "$out/trace.err"; }
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 tools/latency_loop.py 1048576 500 > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; }
done
python3 tools/pmc_summary.py "$out" > "$out/summary.txt" 2>&1
f=$(ls -t $out/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/rocprof_summary.py "$f" > "$out/latency_kernel_stats.txt"
cat "$out/latency_kernel_stats.txt" 2>/dev/null | head -8
cat "$out/latency_under_prof.json"
