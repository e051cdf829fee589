set -e
cd $GRAFT_REPO_ROOT
V=levelsetfusion-python_amd/lib/variants
python tools/ab_state_kernel.py --sizes 256,512 base= aux1=$V/aux1.so aux16=$V/aux16.so aux17=$V/aux17.so aux2=$V/aux2.so > gpurun_out/r02_ab_aux.log 2>&1
cut -c1-110 gpurun_out/r02_ab_aux.log
