set -e
cd $GRAFT_REPO_ROOT
PAIRS=1 python tools/ab_state_kernel.py --sizes 256,512 pairs2= pairs3=levelsetfusion-python_amd/lib/variants/pair3.so > gpurun_out/r02_ab_pairs.log 2>&1
cut -c1-120 gpurun_out/r02_ab_pairs.log
