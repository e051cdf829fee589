set -e
cd $GRAFT_REPO_ROOT
for t in 1024 768 512 256; do echo "threads $t"; LSF_LIST_THREADS=$t python tools/ab_state_kernel.py --sizes 256 packed= ; done > gpurun_out/r02_scale.log 2>&1
for b in 256 192 128 64; do echo "blocks $b"; LSF_LIST_BLOCKS=$b python tools/ab_state_kernel.py --sizes 256 packed= ; done >> gpurun_out/r02_scale.log 2>&1
cat gpurun_out/r02_scale.log
