cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_awq_gpu.py -m gpu -q 2>&1 | tail -3
(python tools/awq_bench.py 16) > gpurun_out/r2be_awq_bench.txt 2>&1
cat gpurun_out/r2be_awq_bench.txt
