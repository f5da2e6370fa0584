cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for fp in 0 1 0 1; do
  if [ $fp = 1 ]; then export CT2X_STREAM_FAKE_PACKED=1; else unset CT2X_STREAM_FAKE_PACKED; fi
  python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>> gpurun_out/r2ao.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fake_packed=$fp', d['value'], d['config']['decode_ms_per_step'], d['config']['prefill_ms_last_call'])"
done > gpurun_out/r2ao_l8_fake_packed.txt 2>&1
cat gpurun_out/r2ao_l8_fake_packed.txt
