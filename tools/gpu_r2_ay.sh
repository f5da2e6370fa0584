cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
# Same box, alternating: the library of this commit vs the library at the start of the session
# (652d115, build/ab/libct2_r2a.so). beam 1 (+ the beam-4 leg bench.py appends) and the Llama shape.
for i in 1 2; do
 for lib in new r2a; do
  if [ $lib = r2a ]; then export CT2X_LIBRARY=$PWD/build/ab/libct2_r2a.so; else unset CT2X_LIBRARY; fi
  python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib beam1', round(d['value']), round(d['ms_per_step'],2), d['config'].get('kernels_per_decode_step'), 'beam4', (d['config'].get('beam4') or {}).get('value'))"
 done
done > gpurun_out/r2ay_same_box.txt 2>&1
for lib in new r2a; do
  if [ $lib = r2a ]; then export CT2X_LIBRARY=$PWD/build/ab/libct2_r2a.so; else unset CT2X_LIBRARY; fi
  python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib l8', round(d['value']), round(d['config']['decode_ms_per_step'],3), 'ms/step', round(d['decode_roofline']['frac'],4))"
done >> gpurun_out/r2ay_same_box.txt 2>&1
cat gpurun_out/r2ay_same_box.txt
