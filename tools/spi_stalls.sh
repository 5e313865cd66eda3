cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name\s+:\s+SPI_RA" | awk '{print $3}' | sort -u > gpurun_out/spi_list.txt
cat gpurun_out/spi_list.txt | tr '\n' ' '
echo
for ctrs in "MeanOccupancyPerCU" "SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_RA_TMP_STALL_CSN SPI_RA_WAVE_SIMD_FULL_CSN" "SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_BAR_CU_FULL_CSN" "SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WVLIM_STALL_CSN SPI_CSN_WAVE SPI_CSN_BUSY"; do
  rm -rf gpurun_out/spi_tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/spi_tmp -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/spi_tmp.log 2>&1 || { echo "FAILED $ctrs"; tail -3 gpurun_out/spi_tmp.log; continue; }
  python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/spi_tmp/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_knn" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k:sum(v)/len(v) for k,v in acc.items()})
PY
done
