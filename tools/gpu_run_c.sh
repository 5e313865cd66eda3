#!/bin/bash
# host-API work: GPU suite, PCIe-inclusive rates, latency report, bench
set -o pipefail
OUT=gpurun_out/r2c
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $OUT/tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -12 $OUT/tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python tools/pcie_inclusive.py > $OUT/pcie_inclusive.json 2> $OUT/pcie_inclusive.err; echo "pcie rc=$?"; cat $OUT/pcie_inclusive.json
timeout -k 10 600 python tools/latency_report.py $OUT/latency.json > $OUT/latency.log 2>&1; echo "latency rc=$?"; tail -3 $OUT/latency.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_uniform.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r2c/bench_uniform.json").read().strip().splitlines()[-1]); print(d["value"], d["extra"])
PY
