cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "quantize_on_load or norm_gemm or gemm or dense" 2>&1 | tail -3
for i in 1 2 3; do
 for q in 1 0; do
  CT2X_QUANT_ON_LOAD=$q python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qol=$q', d['value'], d['ms_per_step'], d['config'].get('kernels_per_decode_step'))"
 done
done > gpurun_out/r2ax_ab.txt 2>&1
cat gpurun_out/r2ax_ab.txt
CT2X_QUANT_ON_LOAD=1 timeout 600 python -m pytest tests/test_translator_gpu.py tests/test_parity_taps_gpu.py -m gpu -q -x 2>&1 | tail -2
