#!/bin/bash
# Round 6: what binds the fused KillingFusion iteration kernel, per variant -- rocprofv3 --pmc passes (one counter group per
# pass, --kernel-trace only besides --pmc) over tools/floor_cases.py for the list walk and the box walk, with and without the
# energy sums, at 256^3 and 512^3; tools/floor_summary.py turns the passes into profiles/r06_floor.txt.
# usage: tools/floor_table.sh <out dir under gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-floor}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for case in "256 list 1" "256 list 0" "256 box 1" "256 box 0" "512 list 1" "512 box 1"; do
  set -- $case
  export N=$1 WALK=$2 ENERGY=$3
  D=$OUT/n${N}_${WALK}_e${ENERGY}
  mkdir -p $D
  python3 $R/tools/floor_cases.py > $D/events.txt 2>&1
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools/floor_cases.py > $D/stats.log 2>&1
  i=0
  for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
               "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVES" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
               "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" \
               "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
               "GRBM_GUI_ACTIVE VALUBusy MemUnitBusy"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $D/p$i -- python3 $R/tools/floor_cases.py > $D/p$i.log 2>&1
    echo "$case pass $i: $(find $D/p$i -name '*counter_collection.csv' | wc -l) file(s)"
  done
done
python3 $R/tools/floor_summary.py $OUT > $OUT/floor.txt 2>&1
tail -40 $OUT/floor.txt
