cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_reftests.py tests/test_pyext.py tests/test_level1_gpu.py -q -s -m gpu > gpurun_out/r2ae_reftests.log 2>&1; grep -v "^$" gpurun_out/r2ae_reftests.log | tail -40
