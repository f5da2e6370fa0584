cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
./build/tools/p_full | cut -c1-60 > gpurun_out/r3/attn128_probe_h.txt 2>&1; cat gpurun_out/r3/attn128_probe_h.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "attention" > gpurun_out/r3/pytest_attn.log 2>&1; tail -3 gpurun_out/r3/pytest_attn.log
timeout 900 python -m pytest tests/test_generator_gpu.py -q -s > gpurun_out/r3/pytest_l3g.log 2>&1; tail -6 gpurun_out/r3/pytest_l3g.log
timeout 600 python tools/bench_l8.py --steps 2 --warmup 1 --no-profile > gpurun_out/r3/l8_d.json 2> gpurun_out/r3/l8_d.err; python -c "
import json;d=json.loads(open('gpurun_out/r3/l8_d.json').read().strip().splitlines()[-1]);print(d['value'],d['config']['prefill_ms_last_call'],d['config']['decode_ms_per_step'])"
