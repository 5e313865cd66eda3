#!/bin/bash
# parity subset + rebuild check (+ counters with "pmc"):  gpurun --timeout 900 -- 'bash tools/gpu_quick.sh <tag> [pmc]'
set -u
tag=${1:-quick}
out=gpurun_out/$tag
mkdir -p "$out"
bash tools/rebuild_check.sh "$tag" tests || exit 1
if [ "${2:-}" = "pmc" ]; then
  bash tools/pmc_rebuild.sh "$out/pmc_rebuild" > "$out/pmc_rebuild.txt" 2>&1 || { echo "pmc failed"; tail -3 "$out/pmc_rebuild.txt"; }
  python3 - "$out/pmc_rebuild/pmc_summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
keys=("GRBM_GUI_ACTIVE","FETCH_SIZE","WRITE_SIZE","SQ_WAVES","SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","SQ_LDS_BANK_CONFLICT","TCC_HIT_sum","TCC_MISS_sum")
for k in sorted(d):
    if k.startswith(("k_sort", "k_fill", "k_codes", "k_bbox")):
        print(k, {c: round(d[k][c]["avg_per_dispatch"]) for c in keys if c in d[k]})
PY
fi
