cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_translator_gpu.py tests/test_generator_gpu.py -m gpu -q -x 2>&1 | tail -4
python bench.py --no-cpu-baseline > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_bench.err; cut -c1-260 gpurun_out/r2h_bench.json; python -c "
import json; d=json.loads(open('gpurun_out/r2h_bench.json').read().strip().splitlines()[-1]); print(d['config']['kernels_per_decode_step'], json.dumps(d['config'].get('beam4'))[:200]); print(json.dumps(d.get('profile'))[:1500])"
CT2X_ROW_FUSION=0 python bench.py --no-cpu-baseline --no-profile 2>/dev/null | cut -c1-200
