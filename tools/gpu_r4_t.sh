cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "cross_attention_16bit" > $O/t_pytest_xattn.log 2>&1; tail -3 $O/t_pytest_xattn.log
timeout 900 python -m pytest tests/test_translator_gpu.py tests/test_whisper_gpu.py -q -x > $O/t_pytest_models.log 2>&1; tail -3 $O/t_pytest_models.log
timeout 900 python tools/bench_nllb.py --steps 2 --warmup 1 > $O/t_nllb.json 2> $O/t_nllb.err; python -c "
import json; d=json.loads(open('$O/t_nllb.json').read().strip().splitlines()[-1]); print('NLLB', d['value'], d['config']['decode_ms_per_step'], d['config']['kernels_per_decode_step'])"
