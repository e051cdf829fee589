#!/bin/bash
# every rocprofv3 table a round commits, in one GPU call (run through gpurun; ~8 minutes):
#   <tag>_256 / <tag>_512   the default workload (BASELINE config 4) at both sizes: --kernel-trace --stats, FETCH_SIZE and
#                           WRITE_SIZE passes, the calibration launches (tools/profile_bench.sh)
#   <tag>_hiertik / _hierfull / _multiframe / _sobolev   --kernel-trace --stats of the other bench workloads
# then, on the build machine: tools/summarize_profile.py gpurun_out/<tag>_256 <tag> ; ... gpurun_out/<tag>_512 <tag>_512 512 ;
#   ... gpurun_out/<tag>_hierfull <tag>_hierfull  (stats only)
# usage: tools/profile_round.sh <tag>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r05}
mkdir -p $R/gpurun_out
BENCH_ARGS="--no-secondary" bash $R/tools/profile_bench.sh ${T}_256 > $R/gpurun_out/${T}_256.log 2>&1
echo "256 done"
BENCH_ARGS="--size 512 --no-secondary" bash $R/tools/profile_bench.sh ${T}_512 > $R/gpurun_out/${T}_512.log 2>&1
echo "512 done"
for w in hier-tik hier-full sobolev; do
  PASSES=stats BENCH_ARGS="--workload $w" bash $R/tools/profile_bench.sh ${T}_$(echo $w | tr -d -) > $R/gpurun_out/${T}_$w.log 2>&1
  echo "$w done"
done
PASSES=stats BENCH_STEPS=1 BENCH_ARGS="--workload multiframe" bash $R/tools/profile_bench.sh ${T}_multiframe > $R/gpurun_out/${T}_multiframe.log 2>&1
echo "multiframe done"
