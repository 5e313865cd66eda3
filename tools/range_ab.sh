#!/bin/bash
# k_range timing for libpcpx.so and the build variants libpcpx_<tag>.so named in AB_TAGS (default: all): bash tools/range_ab.sh [reps]
tags=${AB_TAGS:-$(ls point-cloud-processing_amd/libpcpx_*.so 2>/dev/null | sed 's/.*libpcpx_\(.*\)\.so/\1/')}
for rnd in $(seq 1 ${AB_ROUNDS:-1}); do
for tag in default $tags; do
  lib=point-cloud-processing_amd/libpcpx_$tag.so
  [ "$tag" = default ] && lib=point-cloud-processing_amd/libpcpx.so
  [ -f "$lib" ] || continue
  PCPX_LIB=$PWD/$lib timeout -k 10 200 python tools/range_loop.py 1e7 ${1:-10} | sed "s|^|$tag: |"
done
done
