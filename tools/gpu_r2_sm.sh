cd $GRAFT_REPO_ROOT
timeout 25 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "softmax" 2>&1 | tail -4
