#!/bin/bash
# parity subset + the four kNN workloads (device resident): python-free summary lines
OUT=gpurun_out/quick
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
for w in uniform_10m_k15 clustered_10m_k15 uniform_10m_k8 uniform_10m_k32_stream; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --workload $w > $OUT/bench_$w.json 2>> $OUT/err.log
python - "$OUT/bench_$w.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["config"]["workload"], d["value"], d["ms_per_step"], d["extra"].get("k_knn_avg_launch_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
