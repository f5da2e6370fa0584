cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
 for lib in new prev; do
  if [ $lib = prev ]; then export CT2X_LIBRARY=$PWD/build/ab/libct2_prev.so; else unset CT2X_LIBRARY; fi
  python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['config'].get('kernels_per_decode_step'), (d['config'].get('beam4') or {}).get('value'))"
 done
done > gpurun_out/r2au_ab.txt 2>&1
cat gpurun_out/r2au_ab.txt
