cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2aj_pytest.log 2>&1; tail -4 gpurun_out/r2aj_pytest.log
python bench.py --no-cpu-baseline > gpurun_out/r2aj_bench.json 2> gpurun_out/r2aj_bench.err
CT2X_QUANT_ON_LOAD=0 python bench.py --no-cpu-baseline --no-l8 --no-profile > gpurun_out/r2aj_bench_noqol.json 2>> gpurun_out/r2aj_bench.err
python - <<'PY'
import json
for f in ('gpurun_out/r2aj_bench.json','gpurun_out/r2aj_bench_noqol.json'):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['config'].get('kernels_per_decode_step'), (d['config'].get('beam4') or {}).get('value'), (d['config'].get('l8') or {}).get('decode_ms_per_step'))
    except Exception as e:
        print(f, 'ERR', e)
PY
hipcc --offload-arch=gfx950 -O3 -x hip tools/gemm_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/gb 2>/dev/null
SH="2048 14336 4096 4096 4096 4096 8192 8192 8192"
(echo "== exact_div"; /tmp/gb $SH; echo "== CT2X_EXACT_DIV=0 (IEEE division sequence)"; CT2X_EXACT_DIV=0 /tmp/gb $SH) > gpurun_out/r2aj_gemm_big.txt 2>&1
cat gpurun_out/r2aj_gemm_big.txt
