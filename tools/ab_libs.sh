#!/bin/bash
# tools/ab_quick.py over the shipped library and every libpcpx_<tag>.so (AB_TAGS), AB_ROUNDS times round-robin; AB_ARGS = its arguments
out=gpurun_out/ablibs
mkdir -p $out
tags=${AB_TAGS:-$(ls point-cloud-processing_amd/libpcpx_*.so 2>/dev/null | sed 's/.*libpcpx_\(.*\)\.so/\1/')}
for rnd in $(seq 1 ${AB_ROUNDS:-1}); do
for tag in default $tags; do
  lib=point-cloud-processing_amd/libpcpx_$tag.so
  [ "$tag" = default ] && lib=point-cloud-processing_amd/libpcpx.so
  [ -f "$lib" ] || continue
  PCPX_LIB=$PWD/$lib timeout -k 10 200 python tools/ab_quick.py ${AB_ARGS:-clustered 1e7 15} >> $out/results.jsonl 2>> $out/err.log || exit 1
  tail -1 $out/results.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['lib'], d['kind'], 'whole', d['whole lpt=0'], d['whole lpt=1'], 'eighth slowest/mean', d['eighth lpt=0']['slowest'], d['eighth lpt=0']['mean'], '| lpt', d['eighth lpt=1']['slowest'], d['eighth lpt=1']['mean'], 'x', d['speedup_8 (best whole / slowest eighth, lpt=1)'], d.get('group ticks/64: mean p99 max'))"
done
done
