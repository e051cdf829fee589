#!/bin/bash
# rocprofv3 --pmc passes over tools/hier_kernel_time.py (hier_iteration_kernel<TIK, UPDATE> at 256^3)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_hier}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for group in "VALUBusy SALUBusy" "OccupancyPercent MemUnitStalled" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "L2CacheHit"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/hier_kernel_time.py > $OUT/p$i.log 2>&1
  echo "pass $i ($group): $(find $OUT/p$i -name '*counter_collection.csv' | wc -l) file(s)"
done
python3 $R/tools/summarize_pmc.py $OUT hier_iteration > $OUT/summary.txt 2>&1
