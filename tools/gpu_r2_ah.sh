cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pyext.py tests/test_reftests.py tests/test_level1_gpu.py -q -m gpu 2>&1 | tail -6
