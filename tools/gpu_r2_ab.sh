cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "attention" 2>&1 | tail -2
for r in 0 1; do
  echo "== NO_XCD_REMAP=$r"
  if [ $r = 1 ]; then export CT2X_ATTN_NO_XCD_REMAP=1; else unset CT2X_ATTN_NO_XCD_REMAP; fi
  python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['decode_ms_per_step'], d['decode_roofline']['frac'])"
done
unset CT2X_ATTN_NO_XCD_REMAP
python tools/bench_l8.py --steps 2 --warmup 1 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in d['profile']['top'][:6]: print('%-48s %6d %8.2f us %5.1f%%' % (k['kernel'], k['launches'], k['avg_us'], 100*k['share']))"
