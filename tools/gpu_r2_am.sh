cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for nt in 0 1 0 1; do
  CT2X_NT_WEIGHTS=$nt python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>> gpurun_out/r2am.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nt=$nt', d['value'], d['config']['decode_ms_per_step'], d['config']['prefill_ms_last_call'])"
done > gpurun_out/r2am_l8_nt.txt 2>&1
cat gpurun_out/r2am_l8_nt.txt
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "gemm or dense" 2>&1 | tail -2
