#!/bin/bash
# rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over tools/pmc_fused.py
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_fused}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for group in "VALUBusy SALUBusy" "VALUUtilization OccupancyPercent" "MemUnitBusy MemUnitStalled" "TA_BUSY_avr TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "L2CacheHit"; do
  i=$((i+1))
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/pmc_fused.py > $OUT/p$i.log 2>&1
  echo "pass $i ($group): $(find $OUT/p$i -name '*counter_collection.csv' | wc -l) file(s)"
done
