set -u
mkdir -p gpurun_out/r04r
python bench.py > gpurun_out/r04r/bench_default.json 2> gpurun_out/r04r/bench_default.err
python bench.py --size 512 --steps 10 --no-secondary > gpurun_out/r04r/bench_512.json 2>/dev/null
python bench.py --data depth --no-secondary --no-cpu-baseline > gpurun_out/r04r/bench_depth_256.json 2>/dev/null
python bench.py --data depth --size 512 --steps 10 --no-secondary --no-cpu-baseline > gpurun_out/r04r/bench_depth_512.json 2>/dev/null
python tools/host_timeline.py 256 > gpurun_out/r04r/host_timeline_256.txt 2>&1
HALO=8 ITERS=50 FIXED_ONLY=1 LB_TIMELINE=1 python tools/slab_nccl_loopback.py 256 > gpurun_out/r04r/loopback_faces.txt 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r04r/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04r/tests.log
tail -3 gpurun_out/r04r/tests.log
