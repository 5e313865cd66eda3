#!/bin/bash
# rocprofv3 counters for the index rebuild (tools/rebuild_loop.py 1e7 6), separate passes; prints the sort / leaf kernels
out=${1:-gpurun_out/pmc_rebuild}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for ctrs in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_GDS" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 tools/rebuild_loop.py ${PMC_N:-1e7} 6 > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; }
done
python3 tools/pmc_summary.py "$out" > /dev/null
python3 - "$out/pmc_summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in sorted(d):
    if k.startswith(("k_sort", "k_fill", "k_codes", "k_bbox", "k_upper")): print(k, json.dumps({c: round(v["avg_per_dispatch"],1) for c,v in d[k].items()}))
PY
