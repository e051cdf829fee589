mkdir -p gpurun_out/r04v
cd /tmp && export TMPDIR=/tmp
export HALO=8 ITERS=50 FIXED_ONLY=1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04v/trace -- python3 $GRAFT_REPO_ROOT/tools/slab_nccl_loopback.py 256 > $GRAFT_REPO_ROOT/gpurun_out/r04v/loopback.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r04v -name "*kernel_trace.csv" | head
