#!/bin/bash
# Hilbert order: whole GPU suite, bench (uniform, clustered, C5), traversal statistics
set -o pipefail
OUT=gpurun_out/r2b
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $OUT/tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_uniform.json 2> $OUT/bench.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload clustered_10m_k15 > $OUT/bench_clustered.json 2>> $OUT/bench.err && \
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload uniform_50m_k32_stream > $OUT/bench_c5.json 2>> $OUT/bench.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload uniform_10m_k8 > $OUT/bench_k8.json 2>> $OUT/bench.err
echo "bench rc=$?"
timeout -k 10 200 python tools/knn_stats.py 1e7 uniform 15 > $OUT/stats_uniform.json 2> $OUT/stats.err
timeout -k 10 200 python tools/knn_stats.py 1e7 clustered 15 > $OUT/stats_clustered.json 2>> $OUT/stats.err
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], d["extra"].get("index_rebuild_ms"), d["extra"].get("config3_range_count_r0.01_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
python - <<'PY'
import json
for k in ("uniform","clustered"):
    try:
        d=json.load(open("gpurun_out/r2b/stats_%s.json"%k)); print(k, d["per_group"])
    except Exception as e: print(k,"ERR",e)
PY
