#!/bin/bash
OUT=gpurun_out/r2h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "radix or rebuild or knn_self_small or bunny_golden or batch_arbitrary" > $OUT/tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -4 $OUT/tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/rebuild_loop.py 1e7 10 > $OUT/rebuild_10m.json 2> $OUT/err.log; cat $OUT/rebuild_10m.json
timeout -k 10 300 python3 tools/rebuild_loop.py 5e7 5 > $OUT/rebuild_50m.json 2>> $OUT/err.log; cat $OUT/rebuild_50m.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/rebuild_loop.py 1e7 10 > $OUT/rebuild_under_prof.json 2>> $OUT/err.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); python3 tools/rocprof_summary.py $f | tee $OUT/rebuild_kernel_stats.txt
