#!/bin/bash
OUT=gpurun_out/r2g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -4 $OUT/tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/rebuild_loop.py 1e7 10 > $OUT/rebuild_10m.json 2> $OUT/err.log; cat $OUT/rebuild_10m.json
timeout -k 10 300 python3 tools/rebuild_loop.py 5e7 5 > $OUT/rebuild_50m.json 2>> $OUT/err.log; cat $OUT/rebuild_50m.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/rebuild_loop.py 1e7 10 > $OUT/rebuild_under_prof.json 2>> $OUT/err.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); python3 tools/rocprof_summary.py $f | tee $OUT/rebuild_kernel_stats.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_uniform.json 2>> $OUT/err.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload uniform_50m_k32_stream > $OUT/bench_c5.json 2>> $OUT/err.log
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["extra"].get("index_rebuild_ms"), d["extra"].get("config3_range_count_r0.01_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
