#!/bin/bash
# A/B of k_knn build variants (GPU box): libpcpx.so and every libpcpx_<tag>.so (AB_TAGS="tag1 tag2", default: all) on the kNN
# workloads (AB_WORKLOADS), AB_ROUNDS times round-robin so that box drift shows
out=gpurun_out/abknn
mkdir -p $out
tags=${AB_TAGS:-$(ls point-cloud-processing_amd/libpcpx_*.so 2>/dev/null | sed 's/.*libpcpx_\(.*\)\.so/\1/')}
workloads=${AB_WORKLOADS:-uniform_10m_k15 clustered_10m_k15 uniform_10m_k8 uniform_10m_k32_stream}
for rnd in $(seq 1 ${AB_ROUNDS:-1}); do
for tag in default $tags; do
  lib=point-cloud-processing_amd/libpcpx_$tag.so
  [ "$tag" = default ] && lib=point-cloud-processing_amd/libpcpx.so
  [ -f "$lib" ] || continue
  export PCPX_LIB=$PWD/$lib
  line="$tag:"
  for w in $workloads; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --workload $w > $out/${tag}_$w.json 2>> $out/err.log || exit 1
    v=$(python -c "import json,sys;print(json.loads(open('$out/${tag}_$w.json').read().strip().splitlines()[-1])['value'])")
    line="$line  $w $v"
  done
  echo "$line"
done
done
