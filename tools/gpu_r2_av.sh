cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_translator_gpu.py tests/test_generator_gpu.py tests/test_parity_taps_gpu.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2 3; do
 for lib in new prev; do
  if [ $lib = prev ]; then export CT2X_LIBRARY=$PWD/build/ab/libct2_prev.so; else unset CT2X_LIBRARY; fi
  python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['config'].get('kernels_per_decode_step'))"
 done
done > gpurun_out/r2av_ab.txt 2>&1
cat gpurun_out/r2av_ab.txt
