cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_l8.py --steps 2 --warmup 1 > gpurun_out/r2t_l8.json 2> gpurun_out/r2t_l8.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2t_l8.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['decode_ms_per_step'], d['decode_roofline']['frac'])
prof=d.get('profile') or d.get('kernels')
for k in prof['top']: print("%-50s %6d launches %8.2f us avg %6.1f%%" % (k['kernel'], k['launches'], k['avg_us'], 100*k['share']))
print('noop', prof.get('noop_bracket_us'))
PY
mkdir -p gpurun_out/prof_r2t
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2t -- python $GRAFT_REPO_ROOT/bench.py --beam 4 --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-l8 > $GRAFT_REPO_ROOT/gpurun_out/r2t_beam4_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2t_rocprof.err)
find gpurun_out/prof_r2t -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r2t -name "*.db" -delete
f=$(find gpurun_out/prof_r2t -name "*kernel_stats.csv" | head -1); head -20 "$f" | cut -c1-60,150-260
