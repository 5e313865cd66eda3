#!/bin/bash
OUT=gpurun_out/r2p
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -4 $OUT/tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $OUT/bench_uniform.json 2> $OUT/err.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --workload clustered_10m_k15 > $OUT/bench_clustered.json 2>> $OUT/err.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --workload uniform_10m_k8 > $OUT/bench_k8.json 2>> $OUT/err.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --workload uniform_10m_k32_stream > $OUT/bench_k32.json 2>> $OUT/err.log
timeout -k 10 200 python tools/knn_stats.py 1e7 uniform 15 > $OUT/stats_uniform.json 2>> $OUT/err.log
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["extra"].get("k_knn_avg_launch_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
python -c "
import json; d=json.load(open('$OUT/stats_uniform.json')); print(d['per_group'])"
