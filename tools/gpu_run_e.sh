#!/bin/bash
OUT=gpurun_out/r2e
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/tests.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python tools/pcie_inclusive.py > $OUT/pcie_inclusive.json 2> $OUT/pcie_inclusive.err; echo "pcie rc=$?"; cat $OUT/pcie_inclusive.json
timeout -k 10 600 python tools/latency_report.py $OUT/latency.json > $OUT/latency.log 2>&1; echo "latency rc=$?"; tail -2 $OUT/latency.log
PCPX_FEW_NO_POLL=1 timeout -k 10 100 /tmp/latency_bench 1048576 3000 > $OUT/latency_nopoll.json 2>&1; cat $OUT/latency_nopoll.json
