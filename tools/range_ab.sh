#!/bin/bash
# k_range timing for libpcpx.so and every libpcpx_h*.so: bash tools/range_ab.sh [reps]
for lib in point-cloud-processing_amd/libpcpx.so point-cloud-processing_amd/libpcpx_h*.so; do
  [ -f "$lib" ] || continue
  PCPX_LIB=$PWD/$lib timeout -k 10 200 python tools/range_loop.py 1e7 ${1:-10} | sed "s|^|$(basename $lib .so): |"
done
