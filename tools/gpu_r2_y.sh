cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/pmc_traffic.py gpurun_out/pmc_r02 > gpurun_out/r2y_pmc_traffic.json 2> gpurun_out/r2y_pmc.err; head -c 1500 gpurun_out/r2y_pmc_traffic.json; tail -3 gpurun_out/r2y_pmc.err
find gpurun_out/pmc_r02 -name "*.csv" -size +2M -delete; find gpurun_out/pmc_r02 -name "*.db" -delete


