#!/bin/bash
# A/B: radix sort on key bits [24,64) (5 passes) vs [32,64) (4 passes): rebuild time and query rates (GPU box)
OUT=gpurun_out/passes
mkdir -p $OUT
for tag in base p4; do
  if [ $tag = base ]; then lib=point-cloud-processing_amd/libpcpx.so; else lib=point-cloud-processing_amd/libpcpx_p4.so; fi
  export PCPX_LIB=$PWD/$lib
  echo "== $tag" | tee -a $OUT/log.txt
  timeout -k 10 200 python tools/rebuild_loop.py 1e7 20 >> $OUT/log.txt 2>&1 || exit 1
  timeout -k 10 200 python tools/rebuild_loop.py 5e7 10 >> $OUT/log.txt 2>&1 || exit 1
  for w in uniform_10m_k15 clustered_10m_k15 uniform_10m_k8 uniform_50m_k32_stream; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --workload $w > $OUT/b_${tag}_$w.json 2>> $OUT/err.log || exit 1
    python - "$OUT/b_${tag}_$w.json" >> $OUT/log.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["config"]["workload"], d["value"], d["ms_per_step"])
PY
  done
done
cat $OUT/log.txt
