cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 100 python -m pytest tests/test_translator_gpu.py tests/test_awq_gpu.py -m gpu -q -x 2>&1 | tail -2
