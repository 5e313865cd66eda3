#!/bin/bash
# A/B of k_knn build variants (GPU box): libpcpx.so and every libpcpx_h*.so on the kNN workloads
out=gpurun_out/abknn
mkdir -p $out
for lib in point-cloud-processing_amd/libpcpx.so point-cloud-processing_amd/libpcpx_h*.so; do
  [ -f "$lib" ] || continue
  tag=$(basename $lib .so)
  export PCPX_LIB=$PWD/$lib
  line="$tag:"
  for w in uniform_10m_k15 clustered_10m_k15 uniform_10m_k8 uniform_10m_k32_stream; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --workload $w > $out/${tag}_$w.json 2>> $out/err.log || exit 1
    v=$(python -c "import json,sys;print(json.loads(open('$out/${tag}_$w.json').read().strip().splitlines()[-1])['value'])")
    line="$line  $w $v"
  done
  echo "$line"
done
