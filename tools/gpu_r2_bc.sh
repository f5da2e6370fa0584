cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_awq_gpu.py -m gpu -q 2>&1 | tail -25
timeout 600 python -m pytest tests/test_level1_gpu.py tests/test_reftests.py -m gpu -q -x 2>&1 | tail -3
