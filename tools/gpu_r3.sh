cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "attention or matmul" > gpurun_out/r3/pytest_attn.log 2>&1; tail -15 gpurun_out/r3/pytest_attn.log
timeout 900 python -m pytest tests/test_generator_gpu.py -q -k "llama3" -s > gpurun_out/r3/pytest_l3g.log 2>&1; tail -15 gpurun_out/r3/pytest_l3g.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3/pytest_b.log 2>&1; tail -8 gpurun_out/r3/pytest_b.log
timeout 600 python bench.py > gpurun_out/r3/bench_b.json 2> gpurun_out/r3/bench_b.err; python -c "
import json;d=json.load(open('gpurun_out/r3/bench_b.json'));print(d['value'],d['config']['l8'])"
