cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/vml_probe.py > gpurun_out/r2b_vml.log 2>&1; cat gpurun_out/r2b_vml.log
timeout 900 python -m pytest tests/test_level1_gpu.py -m gpu -q -s -x > gpurun_out/r2b_level1.log 2>&1; grep -E "Level-1|passed|failed|Error|error" gpurun_out/r2b_level1.log | cut -c1-400 | tail -20
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attention or softmax or norm" > gpurun_out/r2b_ops.log 2>&1; tail -5 gpurun_out/r2b_ops.log
timeout 900 python -m pytest tests/test_parity_taps_gpu.py -m gpu -q -s > gpurun_out/r2b_taps.log 2>&1; grep -E "identical|passed|failed" gpurun_out/r2b_taps.log | cut -c1-500 | tail -12
