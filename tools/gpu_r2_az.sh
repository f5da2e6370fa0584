cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_translator_gpu.py -m gpu -q -x -k "attention or tb or greedy" 2>&1 | tail -2
CT2X_ATTN_SPEC_K=half timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_translator_gpu.py -m gpu -q -x -k "attention or tb or greedy" 2>&1 | tail -2
for i in 1 2 3; do
 for m in 1 half 0; do
  CT2X_ATTN_SPEC_K=$m python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spec=$m', round(d['value']), round(d['ms_per_step'],2))"
 done
done > gpurun_out/r2az_spec.txt 2>&1
cat gpurun_out/r2az_spec.txt
