set -e
cd $GRAFT_REPO_ROOT
for r in 1 2; do
LSF_HIP_LIBRARY=$GRAFT_REPO_ROOT/levelsetfusion-python_amd/lib/variants/head.so python tools/filter_times.py 256 7 30
python tools/filter_times.py 256 7 30
done
python tools/filter_times.py 512 7 10
LSF_HIP_LIBRARY=$GRAFT_REPO_ROOT/levelsetfusion-python_amd/lib/variants/head.so python tools/filter_times.py 512 7 10
