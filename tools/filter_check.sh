#!/bin/bash
# filter parity tests + per-kernel profile of the filter bench (GPU box)
out=gpurun_out/filt
mkdir -p "$out"
timeout -k 10 500 python -m pytest tests/test_gpu_filters.py -m gpu -x -q -s > $out/tests.log 2>&1; tail -4 $out/tests.log
bash tools/filter_profile.sh
cat $out/bench_under_prof.json
