#!/bin/bash
# the kNN workloads, device resident, no side measurements: one summary line each.  usage: bash tools/bench_workloads.sh <outdir> [workloads...]
out=${1:-gpurun_out/bench}; shift
mkdir -p "$out"
ws=${@:-uniform_10m_k15 clustered_10m_k15 uniform_10m_k8 uniform_10m_k32_stream}
for w in $ws; do
  timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --workload $w > "$out/bench_$w.json" 2>> "$out/err.log" || { echo "$w failed"; tail -3 "$out/err.log"; exit 1; }
  python - "$out/bench_$w.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["config"]["workload"], d["value"], "Mq/s", d["ms_per_step"], "ms/step; k_knn", d["extra"].get("k_knn_avg_launch_ms"), "ms")
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
