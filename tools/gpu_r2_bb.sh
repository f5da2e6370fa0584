cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_stream_engine_gpu.py tests/test_generator_gpu.py tests/test_ops_gpu.py -m gpu -q -x -k "stream or generator or norm or quantize" 2>&1 | tail -2
run() { python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>> gpurun_out/r2bb.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['config']['decode_ms_per_step'],3))"; }
(CT2X_WEIGHT_PREFETCH=1 run "prefetch=1 (25,17,36,18 MB, 192 wgs)"
 CT2X_WEIGHT_PREFETCH=0 run "prefetch=0"
 CT2X_WEIGHT_PREFETCH=1 run "prefetch=1 again"
 CT2X_WEIGHT_PREFETCH=0 run "prefetch=0 again"
 CT2X_PREFETCH_MB=25,17,0,0 run "mb 25,17,0,0"
 CT2X_PREFETCH_MB=12,8,18,9 run "mb 12,8,18,9"
 CT2X_PREFETCH_WGS=96 run "wgs 96"
 CT2X_PREFETCH_MB=25,17,60,30 run "mb 25,17,60,30") > gpurun_out/r2bb_prefetch.txt 2>&1
cat gpurun_out/r2bb_prefetch.txt; tail -2 gpurun_out/r2bb.err
