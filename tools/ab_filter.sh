#!/bin/bash
# A/B of filter kernel variants (GPU box): per-kernel averages of tools/filter_bench.py under rocprofv3 for each libpcpx_f*.so
out=gpurun_out/filt_ab
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in point-cloud-processing_amd/libpcpx.so point-cloud-processing_amd/libpcpx_f*.so; do
  [ -f "$lib" ] || continue
  tag=$(basename $lib .so)
  echo "== $tag"
  export PCPX_LIB=$PWD/$lib
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$tag" -- python3 tools/filter_bench.py 1e7 1 > "$out/$tag.json" 2> "$out/$tag.err" || { tail -5 "$out/$tag.err"; exit 1; }
  f=$(ls -t $out/$tag/*/*kernel_stats.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_range_accumulate" in r["Name"]:
        nm=r["Name"].split("namespace)::")[-1][:24]
        print("   %-26s calls %3s avg_us %9.1f" % (nm, r["Calls"], float(r["AverageNs"])/1e3))
PY
done
