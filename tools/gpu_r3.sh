cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 400 python -m pytest tests/test_tensor_parallel_gpu.py -q -x > gpurun_out/r3/pytest_tp.log 2>&1; tail -5 gpurun_out/r3/pytest_tp.log
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_tensor_parallel_gpu.py > gpurun_out/r3/pytest_full.log 2>&1; tail -5 gpurun_out/r3/pytest_full.log
( time timeout 900 python bench.py ) > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err; tail -3 gpurun_out/r3/bench_default.err; cat gpurun_out/r3/bench_default.json
