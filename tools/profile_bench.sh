#!/bin/bash
# rocprofv3 passes behind the roofline numbers (run on the GPU box through gpurun):
#   1. --kernel-trace --stats of the default bench workload (per-kernel durations)
#   2. --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE (separate passes: TCC slots), same command
#   4. the same two counters on launches of known traffic (tools/pmc_calibrate.py)
# PMC passes use only --kernel-trace besides --pmc (no sys/hip/hsa trace domains).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# BENCH_ARGS: extra bench.py flags (e.g. "--size 512", "--workload hier-full"); PASSES: "stats" alone skips the PMC passes
BENCH="python3 $R/bench.py --steps ${BENCH_STEPS:-3} --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_stats.log 2>&1
if [ "${PASSES:-all}" = "stats" ]; then find $OUT -name "*kernel_stats.csv" | head; tail -2 $OUT/bench_stats.log; exit 0; fi
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/bench_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_fetch -- python3 $R/tools/pmc_calibrate.py > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_write -- python3 $R/tools/pmc_calibrate.py > $OUT/cal_write.log 2>&1
find $OUT -name "*.csv" | head -40
tail -2 $OUT/bench_stats.log
