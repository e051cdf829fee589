set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_hooks.py tests/test_gpu_multiframe.py tests/test_gpu_bench_contract.py tests/test_gpu_parity.py -x -q -k "hook or multiframe or multipair or full_size_512 or bench or workload or convolution" > gpurun_out/r02_mf_tests.log 2>&1 || { tail -60 gpurun_out/r02_mf_tests.log | cut -c1-200; exit 1; }
tail -3 gpurun_out/r02_mf_tests.log
python bench.py --workload multiframe --steps 2 --warmup 1 > gpurun_out/r02_bench_multiframe.json 2> gpurun_out/r02_bench_multiframe.err || { tail gpurun_out/r02_bench_multiframe.err; exit 1; }
cat gpurun_out/r02_bench_multiframe.json
