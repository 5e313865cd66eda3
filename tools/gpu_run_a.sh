#!/bin/bash
# first GPU call of round 2: whole GPU suite, PCIe rate microbenchmark, default bench
set -o pipefail
OUT=gpurun_out/r2a
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $OUT/tests.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a $OUT/tests.log
tail -5 $OUT/tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
hipcc -O2 --offload-arch=gfx950 tools/pcie_rate.hip -o /tmp/pcie_rate -pthread && timeout -k 10 120 /tmp/pcie_rate > $OUT/pcie_rate.json 2> $OUT/pcie_rate.err
echo "pcie rc=$?"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
tail -c 1500 $OUT/bench.json
