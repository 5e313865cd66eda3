#!/bin/bash
# A/B of libpcpx build variants on chosen workloads (GPU box): bash tools/ab_one.sh "<workloads>" [rounds]
ws=${1:-uniform_10m_k15}; rounds=${2:-1}
out=gpurun_out/abone; mkdir -p $out
for r in $(seq 1 $rounds); do
for lib in point-cloud-processing_amd/libpcpx.so point-cloud-processing_amd/libpcpx_h*.so; do
  [ -f "$lib" ] || continue
  tag=$(basename $lib .so)
  export PCPX_LIB=$PWD/$lib
  line="$tag:"
  for w in $ws; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --workload $w > $out/${tag}_$w.json 2>> $out/err.log || exit 1
    v=$(python -c "import json,sys;print(json.loads(open('$out/${tag}_$w.json').read().strip().splitlines()[-1])['value'])")
    line="$line  $w $v"
  done
  echo "$line"
done
done
