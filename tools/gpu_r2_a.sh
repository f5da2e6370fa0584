cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "layer_norm or rms_norm or softmax or attention or norm_quantize" > gpurun_out/r2a_ops.log 2>&1; tail -15 gpurun_out/r2a_ops.log
timeout 900 python -m pytest tests/test_parity_taps_gpu.py -m gpu -q -s > gpurun_out/r2a_taps.log 2>&1; grep -E "identical|passed|failed|Error|error" gpurun_out/r2a_taps.log | cut -c1-600 | tail -30
timeout 900 python -m pytest tests/test_translator_gpu.py -m gpu -q -s > gpurun_out/r2a_translator.log 2>&1; grep -E "identical|flip|agree|passed|failed|max \|" gpurun_out/r2a_translator.log | cut -c1-300 | tail -30
python bench.py --no-cpu-baseline > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err; cut -c1-400 gpurun_out/r2a_bench.json
