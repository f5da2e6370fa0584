cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_tensor_parallel_gpu.py tests/test_host.py tests/test_generator_gpu.py -q -x > gpurun_out/r2u_tp.log 2>&1; tail -5 gpurun_out/r2u_tp.log
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "random_sample" > gpurun_out/r2u_ops.log 2>&1; tail -5 gpurun_out/r2u_ops.log
