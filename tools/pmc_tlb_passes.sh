#!/bin/bash
# rocprofv3 --pmc passes over the vector L1's address-translation (UTCL1) and stall counters of the fused list kernel
# (tools/pmc_fused.py at N = 256 and N = 512): does a 512^3 launch -- 2 GiB per state, 4 MiB between the z - 1 / z / z + 1
# rows of a voxel -- pay for translations where a 256^3 launch does not?  One group per pass, --kernel-trace only besides --pmc.
# usage: tools/pmc_tlb_passes.sh <out dir under gpurun_out> "<N list>"
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_tlb}
SIZES=${2:-"256 512"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in $SIZES; do
  export N=$n
  mkdir -p $OUT/n$n
  i=0
  for group in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
               "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
               "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" \
               "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" \
               "TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum"; do
    i=$((i+1))
    timeout -k 10 180 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/n$n/p$i -- python3 $R/tools/pmc_fused.py > $OUT/n$n/p$i.log 2>&1
    echo "N=$n pass $i ($group): $(find $OUT/n$n/p$i -name '*counter_collection.csv' | wc -l) file(s)"
  done
  python3 $R/tools/summarize_pmc.py $OUT/n$n > $OUT/summary_$n.txt 2>&1
done
echo done
