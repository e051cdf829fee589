set -e
cd $GRAFT_REPO_ROOT
bash tools/pmc_passes.sh r02a_pmc > gpurun_out/r02a_pmc.log 2>&1 || true
tail -3 gpurun_out/r02a_pmc.log
