#!/bin/bash
OUT=gpurun_out/r2l
mkdir -p $OUT
for g in 1 3 4 6 8; do echo "min groups per wave $g"; PCPX_MIN_GROUPS_PER_WAVE=$g timeout -k 10 300 python tools/shard_rate.py 2>>$OUT/err.log | tee $OUT/shard_$g.json; done
