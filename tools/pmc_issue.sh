#!/bin/bash
# Counters that say what a query kernel waits for (instruction fetch, scalar cache, scalar/LDS pipes): separate passes.
# usage: tools/pmc_issue.sh <outdir> [bench args...]
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for ctrs in \
  "SQ_INSTS SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES" \
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_BUSY_CYCLES" \
  "SQC_TC_REQ SQC_TC_STALL SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ" \
  "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; }
done
python3 tools/pmc_summary.py "$out" > "$out/summary.txt" 2>&1
python3 - "$out/pmc_summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in sorted(d):
    if k.startswith(("k_knn","k_range")):
        print(k)
        for c,x in sorted(d[k].items()): print("   %-28s %16.0f"%(c,x["avg_per_dispatch"]))
PY
