#!/bin/bash
OUT=gpurun_out/r2n
mkdir -p $OUT
V="base: by8:DPCPX_COMPACT_BY8_32=1 by8w5:DPCPX_COMPACT_BY8_32=1,DPCPX_MINW32=5 buf12:DPCPX_BUF32=12 buf14:DPCPX_BUF32=14 by8buf12:DPCPX_COMPACT_BY8_32=1,DPCPX_BUF32=12 w5:DPCPX_MINW32=5"
python tools/ab_variants.py build $V > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
echo built
AB_ROUNDS=2 AB_STEPS=5 AB_ARGS="--workload uniform_10m_k32_stream" python tools/ab_variants.py run $V > $OUT/ab_k32.log 2>&1; cat $OUT/ab_k32.log
