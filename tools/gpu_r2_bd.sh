cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(python tools/awq_bench.py 16; python tools/awq_bench.py 1) > gpurun_out/r2bd_awq_bench.txt 2>&1
cat gpurun_out/r2bd_awq_bench.txt
