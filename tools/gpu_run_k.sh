#!/bin/bash
# re-tune the kNN kernel's knobs under the Hilbert order; shard timing
OUT=gpurun_out/r2k
mkdir -p $OUT
V="base: seed0:DPCPX_SEED_EXTRA=0 seed1:DPCPX_SEED_EXTRA=1 seed3:DPCPX_SEED_EXTRA=3 seed4:DPCPX_SEED_EXTRA=4 cap125:DPCPX_CAP_MULT=1.25f cap15:DPCPX_CAP_MULT=1.5f cap175:DPCPX_CAP_MULT=1.75f cap2:DPCPX_CAP_MULT=2.0f buf9:DPCPX_BUF16=9 buf11:DPCPX_BUF16=11 buf12:DPCPX_BUF16=12"
python tools/ab_variants.py build $V > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
AB_ROUNDS=2 AB_STEPS=8 python tools/ab_variants.py run $V > $OUT/ab_uniform.log 2>&1; cat $OUT/ab_uniform.log
AB_ROUNDS=1 AB_STEPS=8 AB_ARGS="--workload clustered_10m_k15" python tools/ab_variants.py run $V > $OUT/ab_clustered.log 2>&1; cat $OUT/ab_clustered.log
timeout -k 10 300 python tools/shard_rate.py > $OUT/shard_rate.json 2>$OUT/err.log; cat $OUT/shard_rate.json
