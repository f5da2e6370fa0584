cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2c_pytest.log 2>&1; tail -15 gpurun_out/r2c_pytest.log
timeout 600 python -m pytest tests/test_translator_gpu.py tests/test_parity_taps_gpu.py -m gpu -q -s > gpurun_out/r2c_parity.log 2>&1; grep -E "identical|flip|agree" gpurun_out/r2c_parity.log | cut -c1-260 | head -40
python bench.py --no-cpu-baseline > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err; cut -c1-300 gpurun_out/r2c_bench.json
