cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm_half_weights or norm" > $O/n_pytest_ops.log 2>&1; tail -3 $O/n_pytest_ops.log
timeout 900 python -m pytest tests/test_translator_gpu.py tests/test_whisper_gpu.py tests/test_generator_gpu.py -q -x > $O/n_pytest_models.log 2>&1; tail -4 $O/n_pytest_models.log
timeout 900 python tools/bench_nllb.py --steps 2 --warmup 1 > $O/n_nllb.json 2> $O/n_nllb.err; python -c "
import json; d=json.loads(open('$O/n_nllb.json').read().strip().splitlines()[-1]); print('NLLB', d['value'], d['config']['decode_ms_per_step'], d['config']['kernels_per_decode_step'])"
timeout 900 python tools/bench_whisper.py --steps 2 --warmup 1 > $O/n_whisper.json 2> $O/n_whisper.err; tail -c 700 $O/n_whisper.json
