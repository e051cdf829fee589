mkdir -p gpurun_out/r04x
for rep in 1 2; do for v in always first; do for s in 1 0; do echo "== LSF_SLAB_VERIFY_FACES=$v LSF_SLAB_SPLIT=$s (rep $rep)" >> gpurun_out/r04x/loopback_ab.txt; LSF_SLAB_VERIFY_FACES=$v LSF_SLAB_SPLIT=$s HALO=8 ITERS=50 FIXED_ONLY=1 python tools/slab_nccl_loopback.py 256 2>&1 | grep "fixed count" | tail -1 >> gpurun_out/r04x/loopback_ab.txt; done; done; done
cat gpurun_out/r04x/loopback_ab.txt
