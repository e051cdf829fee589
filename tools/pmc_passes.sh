#!/bin/bash
# rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over tools/pmc_fused.py
# (a pass with the TA_BUSY / TA_*_STALLED counters hung on this pool and is not part of the list)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_fused}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for group in "VALUBusy SALUBusy" "VALUUtilization OccupancyPercent" "MemUnitBusy MemUnitStalled" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "L2CacheHit GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/pmc_fused.py > $OUT/p$i.log 2>&1
  echo "pass $i ($group): $(find $OUT/p$i -name '*counter_collection.csv' | wc -l) file(s)"
done
python3 $R/tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
