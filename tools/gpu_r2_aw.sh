cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_translator_gpu.py tests/test_stream_engine_gpu.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do
 for lib in new prev; do
  if [ $lib = prev ]; then export CT2X_LIBRARY=$PWD/build/ab/libct2_prev.so; else unset CT2X_LIBRARY; fi
  python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['config'].get('kernels_per_decode_step'))"
 done
done > gpurun_out/r2aw_ab.txt 2>&1
cat gpurun_out/r2aw_ab.txt
for lib in new prev; do
  if [ $lib = prev ]; then export CT2X_LIBRARY=$PWD/build/ab/libct2_prev.so; else unset CT2X_LIBRARY; fi
  python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l8 $lib', d['value'], d['config']['decode_ms_per_step'])"
done
