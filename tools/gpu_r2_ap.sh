cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_stream_engine_gpu.py -q -m gpu -x -k "gemm or dense or stream" 2>&1 | tail -5
for sp in 1 0 1 0; do
  CT2X_STREAM_PACKED=$sp python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>> gpurun_out/r2ap.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('packed=$sp', d['value'], d['config']['decode_ms_per_step'], d['config']['prefill_ms_last_call'], d['config']['model_load_s'])"
done > gpurun_out/r2ap_l8_packed.txt 2>&1
cat gpurun_out/r2ap_l8_packed.txt; tail -3 gpurun_out/r2ap.err
