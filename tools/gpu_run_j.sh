#!/bin/bash
OUT=gpurun_out/r2j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -15 $OUT/tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
