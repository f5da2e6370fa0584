cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout 300 python -m pytest tests/test_tensor_parallel_gpu.py -q -s -x -k "one_shot" > gpurun_out/r3/pytest_tp.log 2>&1; tail -6 gpurun_out/r3/pytest_tp.log
timeout 900 python -m pytest tests/test_translator_gpu.py tests/test_ops_gpu.py tests/test_generator_gpu.py -q -x > gpurun_out/r3/pytest_c.log 2>&1; tail -4 gpurun_out/r3/pytest_c.log
timeout 500 python tools/bench_nllb.py > gpurun_out/r3/nllb_c.json 2> gpurun_out/r3/nllb_c.err; python -c "
import json;d=json.loads(open('gpurun_out/r3/nllb_c.json').read().strip().splitlines()[-1]);print(d['value'],d['config']['decode_ms_per_step'],d['config']['encode_ms_last_call'])"
