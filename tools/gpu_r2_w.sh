cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_generator_gpu.py -q -s > gpurun_out/r2w_gen.log 2>&1; grep -n "int8\|identical\|passed\|failed\|FAILED\|Error" gpurun_out/r2w_gen.log | head -60
