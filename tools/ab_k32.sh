#!/bin/bash
# A/B of the k <= 32 kernel variants (GPU box): parity with the variant library first, then the k = 32 workloads
out=gpurun_out/abk32
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -2 $out/tests.log
[ $rc -eq 0 ] || exit $rc
for lib in point-cloud-processing_amd/libpcpx.so point-cloud-processing_amd/libpcpx_hk32*.so; do
  export PCPX_LIB=$PWD/$lib
  line="$(basename $lib .so):"
  for w in uniform_10m_k32_stream uniform_50m_k32_stream; do
    timeout -k 10 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --workload $w > $out/$(basename $lib .so)_$w.json 2>> $out/err.log || exit 1
    v=$(python -c "import json;d=json.loads(open('$out/$(basename $lib .so)_$w.json').read().strip().splitlines()[-1]);print(d['value'], d['extra'].get('k_knn_avg_launch_ms'))")
    line="$line  $w $v"
  done
  echo "$line"
done
