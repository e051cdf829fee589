mkdir -p gpurun_out/r04y
HALO=8 ITERS=50 FIXED_ONLY=1 LB_TIMELINE=1 python tools/slab_nccl_loopback.py 256 > gpurun_out/r04y/loopback_faces.txt 2>&1
HALO=8 ITERS=50 FIXED_ONLY=1 PATTERN=centered python tools/slab_nccl_loopback.py 256 > gpurun_out/r04y/loopback_centered.txt 2>&1
grep "fixed count\|single GPU" gpurun_out/r04y/loopback_faces.txt gpurun_out/r04y/loopback_centered.txt
