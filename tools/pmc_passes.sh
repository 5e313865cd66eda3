#!/bin/bash
# Collects rocprofv3 counters for bench.py in separate passes (gfx950: SQ 8 slots, TCC 4: FETCH_SIZE=3, WRITE_SIZE=2).
# usage: tools/pmc_passes.sh <outdir> [bench args...]
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
PASSES=${PMC_PASSES:-5}
for ctrs in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS SQ_INSTS_BRANCH" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" ; do
  i=$((i+1)); [ $i -gt $PASSES ] && break
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 bench.py --no-cpu-baseline --no-extra "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; }
done
python3 tools/pmc_summary.py "$out"
