set -u
mkdir -p gpurun_out/r04q
BENCH_ARGS="--no-secondary" bash tools/profile_bench.sh r04q/prof256 > gpurun_out/r04q/prof256.log 2>&1
BENCH_ARGS="--no-secondary --size 512" bash tools/profile_bench.sh r04q/prof512 > gpurun_out/r04q/prof512.log 2>&1
python bench.py > gpurun_out/r04q/bench_default.json 2> gpurun_out/r04q/bench_default.err
python bench.py --size 512 --steps 10 --no-secondary > gpurun_out/r04q/bench_512.json 2>/dev/null
python bench.py --data depth --no-secondary --no-cpu-baseline > gpurun_out/r04q/bench_depth_256.json 2>/dev/null
python bench.py --data depth --size 512 --steps 10 --no-secondary --no-cpu-baseline > gpurun_out/r04q/bench_depth_512.json 2>/dev/null
python tools/host_timeline.py 256 > gpurun_out/r04q/host_timeline_256.txt 2>&1
echo done
