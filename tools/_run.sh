set -e
cd $GRAFT_REPO_ROOT
V=levelsetfusion-python_amd/lib/variants
python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r02_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r02_gpu_tests.log
python bench.py --no-cpu-baseline > gpurun_out/r02_bench_b.json 2> gpurun_out/r02_bench_b.err || { tail gpurun_out/r02_bench_b.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_bench_b.json'))
print(d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline_dense_walk']['frac'], d['roofline_dense_walk']['kernel_ms'])
PY
python tools/ab_state_kernel.py --sizes 256,512 new= nodefer=$V/nodefer.so > gpurun_out/r02_ab7.log 2>&1
cut -c1-100 gpurun_out/r02_ab7.log
