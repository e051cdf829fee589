mkdir -p gpurun_out/r04w
python -m pytest tests/test_gpu_merge_runs.py tests/test_gpu_slab_two_ranks.py tests/test_gpu_slab_many_ranks.py tests/test_gpu_slab_y.py -x -q -m gpu > gpurun_out/r04w/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04w/tests.log
for rep in 1 2; do HALO=8 ITERS=50 FIXED_ONLY=1 python tools/slab_nccl_loopback.py 256 2>&1 | grep "fixed count\|compact" >> gpurun_out/r04w/loopback.txt; done
cd /tmp && export TMPDIR=/tmp HALO=8 ITERS=50 FIXED_ONLY=1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04w/trace -- python3 $GRAFT_REPO_ROOT/tools/slab_nccl_loopback.py 256 > $GRAFT_REPO_ROOT/gpurun_out/r04w/loopback_traced.log 2>&1
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/r04w/tests.log; cat gpurun_out/r04w/loopback.txt
