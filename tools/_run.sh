set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1; do
LSF_HIER_MAX3=$m rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02_prof_max$m -- python3 $R/bench.py --workload hier-full --no-cpu-baseline --steps 2 --warmup 1 > $R/gpurun_out/r02_prof_hf.log 2>&1
echo "max3=$m"; grep -o "ms_per_step\": [0-9.]*" $R/gpurun_out/r02_prof_hf.log || true
python3 $R/tools/trace_totals.py $R/gpurun_out/r02_prof_max$m 8
done
