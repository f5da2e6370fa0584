cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_stream_engine_gpu.py tests/test_generator_gpu.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do
 for pf in 1 0; do
  CT2X_WEIGHT_PREFETCH=$pf python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>> gpurun_out/r2ba.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefetch=$pf', round(d['value']), round(d['config']['decode_ms_per_step'],3), d['config']['kernels_per_decode_step'])"
 done
done > gpurun_out/r2ba_prefetch.txt 2>&1
CT2X_PREFETCH_MB=25,17,60,30 python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>> gpurun_out/r2ba.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefetch mb 25,17,60,30', round(d['value']), round(d['config']['decode_ms_per_step'],3))" >> gpurun_out/r2ba_prefetch.txt
CT2X_PREFETCH_MB=25,17,0,0 python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>> gpurun_out/r2ba.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefetch mb 25,17,0,0', round(d['value']), round(d['config']['decode_ms_per_step'],3))" >> gpurun_out/r2ba_prefetch.txt
cat gpurun_out/r2ba_prefetch.txt; tail -3 gpurun_out/r2ba.err
