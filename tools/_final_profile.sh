#!/bin/bash
BENCH_ARGS="--no-secondary" bash tools/profile_bench.sh r04Z/prof256 > gpurun_out/r04Z_prof256.log 2>&1
BENCH_ARGS="--no-secondary --size 512" bash tools/profile_bench.sh r04Z/prof512 > gpurun_out/r04Z_prof512.log 2>&1
BENCH_ARGS="--no-secondary --workload sobolev" bash tools/profile_bench.sh r04Z/sobolev > gpurun_out/r04Z_sobolev.log 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r04Z_tests.log 2>&1
tail -n 3 gpurun_out/r04Z_tests.log
