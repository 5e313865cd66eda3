#!/bin/bash
# MeanOccupancyPerCU of k_knn for variant libs: tools/occupancy_probe.sh name1 name2 ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf gpurun_out/occ_tmp
  PCPX_LIB=$GRAFT_REPO_ROOT/point-cloud-processing_amd/libpcpx_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --pmc MeanOccupancyPerCU --output-format csv -d gpurun_out/occ_tmp -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/occ_tmp.log 2>&1
  python3 - "$v" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list); dur=[]
for f in glob.glob("gpurun_out/occ_tmp/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_knn" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/occ_tmp/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_knn" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
print(sys.argv[1], {k:round(sum(v)/len(v),2) for k,v in acc.items()}, "ms", round(sum(dur)/max(1,len(dur)),3))
PY
done
