cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "big or dense_fused or gemm_s8_exact" 2>&1 | tail -2
python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l8', round(d['value']), round(d['config']['decode_ms_per_step'],3), 'prefill ms', round(d['config']['prefill_ms_last_call'],1))"
