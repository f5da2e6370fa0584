cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_translator_gpu.py tests/test_generator_gpu.py -q 2>&1 | tail -8
