#!/bin/bash
# rocprofv3 kernel stats + counters for k_range (configs[2]: 10 M radius counts at r = 0.01), separate passes
out=${1:-gpurun_out/pmc_range}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 tools/range_loop.py 1e7 5 > "$out/range_under_prof.json" 2> "$out/err.log" || { echo "trace failed"; tail -3 "$out/err.log"; }
f=$(find "$out/trace" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python3 tools/rocprof_summary.py "$f" | head -5 | tee "$out/range_kernel_stats.txt"
i=0
for ctrs in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 tools/range_loop.py 1e7 3 > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; }
done
python3 tools/pmc_summary.py "$out" > /dev/null
python3 - "$out/pmc_summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in sorted(d):
    if k.startswith("k_range"): print(k, json.dumps({c: round(v["avg_per_dispatch"],1) for c,v in d[k].items()}))
PY
