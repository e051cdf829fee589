mkdir -p gpurun_out/r04s
for rep in 1 2; do for at in plan first; do echo "== LSF_SLAB_FACE_CHECK_AT=$at (rep $rep)" >> gpurun_out/r04s/loopback_ab.txt; LSF_SLAB_FACE_CHECK_AT=$at HALO=8 ITERS=50 FIXED_ONLY=1 python tools/slab_nccl_loopback.py 256 2>&1 | grep "fixed count" >> gpurun_out/r04s/loopback_ab.txt; done; done
python bench.py > gpurun_out/r04s/bench_default.json 2> gpurun_out/r04s/bench_default.err
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r04s/bench_default2.json 2>/dev/null
cat gpurun_out/r04s/loopback_ab.txt
