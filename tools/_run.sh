set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_slab_many_ranks.py tests/test_gpu_slab_two_ranks.py tests/test_gpu_hooks.py -x -q > gpurun_out/r02_slab_tests.log 2>&1 || { tail -60 gpurun_out/r02_slab_tests.log | cut -c1-220; exit 1; }
tail -3 gpurun_out/r02_slab_tests.log
