#!/bin/bash
OUT=gpurun_out/r2f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 tools/rebuild_loop.py 1e7 10 > $OUT/rebuild_10m.json 2> $OUT/err.log; cat $OUT/rebuild_10m.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/rebuild_loop.py 1e7 10 > $OUT/rebuild_under_prof.json 2>> $OUT/err.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); python3 tools/rocprof_summary.py $f | tee $OUT/rebuild_kernel_stats.txt
