#!/bin/bash
OUT=gpurun_out/r2m
mkdir -p $OUT
V="base: noeps:DPCPX_EXP_NOEPS nowrite:DPCPX_EXP_NOWRITE distonly:DPCPX_EXP_DISTONLY nopos:DPCPX_EXP_NOPOS"
python tools/ab_variants.py build $V > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
AB_ROUNDS=2 AB_STEPS=8 python tools/ab_variants.py run $V > $OUT/ab.log 2>&1; cat $OUT/ab.log
