#!/bin/bash
# One GPU-box visit: the whole -m gpu suite, the rebuild check (timings, kernel stats, variants), the rebuild's counters.
#   gpurun --timeout 1200 -- 'bash tools/gpu_round.sh <tag>'
set -u
tag=${1:-round}
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$out/gpu_tests.log" 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 "$out/gpu_tests.log"
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/rebuild_check.sh "$tag" notests || exit 1
bash tools/pmc_rebuild.sh "$out/pmc_rebuild" > "$out/pmc_rebuild.txt" 2>&1 || { echo "pmc failed"; tail -3 "$out/pmc_rebuild.txt"; }
cat "$out/pmc_rebuild.txt" | cut -c1-900
