#!/bin/bash
OUT=gpurun_out/r2d
mkdir -p $OUT
for mode in none nocopy nocompute; do
  PCPX_DEBUG_PIPE=$mode timeout -k 10 200 python tools/pcie_inclusive.py > $OUT/pipe_$mode.json 2>> $OUT/err.log; echo "$mode rc=$?"
  python -c "
import json;d=json.load(open('$OUT/pipe_$mode.json'));print('$mode',{k:v['ms'] for k,v in d.items() if isinstance(v,dict)})"
done
