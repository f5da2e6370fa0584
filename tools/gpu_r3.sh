cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_tensor_parallel_gpu.py -q -s -x > gpurun_out/r3/pytest_tp.log 2>&1; tail -25 gpurun_out/r3/pytest_tp.log
timeout 900 python bench.py --no-cpu-baseline --no-l8 > gpurun_out/r3/bench_c.json 2> gpurun_out/r3/bench_c.err; python -c "
import json;d=json.load(open('gpurun_out/r3/bench_c.json'));print(d['value']);print(d['config'].get('tp_local'));print(d['config'].get('prefill_gemm'))"; tail -3 gpurun_out/r3/bench_c.err
