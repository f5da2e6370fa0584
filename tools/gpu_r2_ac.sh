cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pyext.py tests/test_level1_gpu.py -q -x > gpurun_out/r2ac_pyext.log 2>&1; tail -15 gpurun_out/r2ac_pyext.log
