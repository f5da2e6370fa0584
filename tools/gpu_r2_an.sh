cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "gemm or dense" 2>&1 | tail -5
for sg in 1 0 1 0; do
  CT2X_STREAM_GEMM=$sg python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>> gpurun_out/r2an.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream=$sg', d['value'], d['config']['decode_ms_per_step'], d['config']['prefill_ms_last_call'])"
done > gpurun_out/r2an_l8_stream.txt 2>&1
cat gpurun_out/r2an_l8_stream.txt; tail -3 gpurun_out/r2an.err
