#!/bin/bash
# final profiles of this build: 256^3, 512^3, sobolev (primary workload only)
BENCH_ARGS="--no-secondary" bash tools/profile_bench.sh r04Q/prof256 > gpurun_out/r04Q_prof256.log 2>&1
BENCH_ARGS="--no-secondary --size 512" bash tools/profile_bench.sh r04Q/prof512 > gpurun_out/r04Q_prof512.log 2>&1
BENCH_ARGS="--no-secondary --workload sobolev" bash tools/profile_bench.sh r04Q/sobolev > gpurun_out/r04Q_sobolev.log 2>&1
python -m pytest tests/test_gpu_sobolev_fused_x.py -x -q -m gpu > gpurun_out/r04Q_tests.log 2>&1
for rep in 1 2; do
for g in 1 0; do
  echo "== LSF_SOBOLEV_KEEP_BUFFERS=$g (rep $rep)" >> gpurun_out/r04Q_keep_ab.txt
  LSF_SOBOLEV_KEEP_BUFFERS=$g python bench.py --no-secondary --workload sobolev 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r04Q_keep_ab.txt
done
done
tail -n 3 gpurun_out/r04Q_tests.log; cat gpurun_out/r04Q_keep_ab.txt
