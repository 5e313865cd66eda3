#!/bin/bash
# shard projections (clustered + uniform) for libpcpx.so and variants
for tag in default $AB_TAGS; do
  lib=point-cloud-processing_amd/libpcpx_$tag.so
  [ "$tag" = default ] && lib=point-cloud-processing_amd/libpcpx.so
  for w in "clustered 1e7 15" "uniform 1e7 15"; do
    PCPX_LIB=$PWD/$lib timeout -k 10 300 python3 tools/shard_rate.py $w > /tmp/sr.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('/tmp/sr.json')); print('$tag', '$w', [(k[6:], d[k].get('speedup_projected'), d[k].get('ms_slowest_rank'), d[k].get('ms_mean_rank')) for k in d if k.startswith('ranks_')])"
  done
done
