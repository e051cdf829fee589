set -u
mkdir -p gpurun_out/r04z
BENCH_ARGS="--no-secondary" bash tools/profile_bench.sh r04z/prof256 > gpurun_out/r04z/prof256.log 2>&1
BENCH_ARGS="--no-secondary --size 512" bash tools/profile_bench.sh r04z/prof512 > gpurun_out/r04z/prof512.log 2>&1
python bench.py > gpurun_out/r04z/bench_default.json 2> gpurun_out/r04z/bench_default.err
python bench.py --size 512 --steps 10 --no-secondary > gpurun_out/r04z/bench_512.json 2>/dev/null
python bench.py --data depth --no-secondary --no-cpu-baseline > gpurun_out/r04z/bench_depth_256.json 2>/dev/null
python bench.py --data depth --size 512 --steps 10 --no-secondary --no-cpu-baseline > gpurun_out/r04z/bench_depth_512.json 2>/dev/null
python tools/host_timeline.py 256 > gpurun_out/r04z/host_timeline_256.txt 2>&1
echo done
BENCH_ARGS="--workload sobolev" bash tools/profile_bench.sh r04z/sobolev > gpurun_out/r04z/sobolev_prof.log 2>&1
python bench.py --workload sobolev --no-cpu-baseline --steps 10 > gpurun_out/r04z/bench_sobolev.json 2>/dev/null
python -m pytest tests -x -q -m gpu > gpurun_out/r04z/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04z/tests.log
tail -3 gpurun_out/r04z/tests.log
