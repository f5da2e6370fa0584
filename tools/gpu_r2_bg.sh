cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_stream_engine_gpu.py -m gpu -q -x -k "stream" 2>&1 | tail -2
for i in 1 2; do
 for lib in new prev; do
  if [ $lib = prev ]; then export CT2X_LIBRARY=$PWD/build/ab/libct2_prev.so; else unset CT2X_LIBRARY; fi
  python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l8 $lib', round(d['value']), round(d['config']['decode_ms_per_step'],3))"
 done
done > gpurun_out/r2bg_l8_nospill.txt 2>&1
cat gpurun_out/r2bg_l8_nospill.txt
