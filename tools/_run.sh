set -e
cd $GRAFT_REPO_ROOT
python tools/_depth_probe.py
LSF_HIP_LIBRARY=$GRAFT_REPO_ROOT/levelsetfusion-python_amd/lib/variants/r01.so python tools/_depth_probe.py
