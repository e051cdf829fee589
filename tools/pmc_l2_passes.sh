#!/bin/bash
# rocprofv3 --pmc passes over the L2 (TCC) and vector-L1 (TCP) counters of the fused list kernel (tools/pmc_fused.py,
# N = 256 and N = 512), one group of at most four TCC counters per pass, --kernel-trace only besides --pmc.
# FETCH_SIZE on gfx950 tallies a 128-byte request at 64 bytes (MI355X_MICROARCH.md, HBM section), so the read bytes are
# taken from the request counters by size: 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B.
# usage: tools/pmc_l2_passes.sh <out dir under gpurun_out> "<N list>"
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_l2}
SIZES=${2:-"256 512"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in $SIZES; do
  export N=$n
  mkdir -p $OUT/n$n
  i=0
  for group in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
               "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
               "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
               "TCC_READ_SECTORS_sum TCC_WRITE_SECTORS_sum TCC_NORMAL_EVICT_sum TCC_NORMAL_WRITEBACK_sum" \
               "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_CACHE_MISS_sum TCP_TCC_WRITE_REQ_sum" \
               "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_LEVEL_sum"; do
    i=$((i+1))
    timeout -k 10 180 rocprofv3 --pmc $group --kernel-trace --output-format csv -d $OUT/n$n/p$i -- python3 $R/tools/pmc_fused.py > $OUT/n$n/p$i.log 2>&1
    echo "N=$n pass $i ($group): $(find $OUT/n$n/p$i -name '*counter_collection.csv' | wc -l) file(s)"
  done
  python3 $R/tools/summarize_pmc.py $OUT/n$n > $OUT/summary_$n.txt 2>&1
done
# the same request-size counters on launches of known traffic (16 bytes per lane streaming reads and writes)
timeout -k 10 180 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/cal/p1 -- python3 $R/tools/pmc_calibrate.py > $OUT/cal_p1.log 2>&1
timeout -k 10 180 rocprofv3 --pmc TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/cal/p2 -- python3 $R/tools/pmc_calibrate.py > $OUT/cal_p2.log 2>&1
python3 $R/tools/summarize_pmc.py $OUT/cal kernel > $OUT/summary_cal.txt 2>&1
echo done
