cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_awq_gpu.py -m gpu -q -x 2>&1 | tail -15
