cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 900 python -m pytest tests/test_compaction_gpu.py -q -x > $O/p_pytest_compaction.log 2>&1; tail -25 $O/p_pytest_compaction.log
