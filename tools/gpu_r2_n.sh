cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_l8.py --steps 2 --warmup 1 > gpurun_out/r2n_l8.json 2> gpurun_out/r2n_l8.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2n_l8.json').read().strip().splitlines()[-1])
print({k:v for k,v in d.items() if k not in ('profile','kernels')})
prof=d.get('profile') or d.get('kernels')
print(json.dumps(prof)[:3000])
PY
mkdir -p gpurun_out/prof_r2n
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2n -- python $GRAFT_REPO_ROOT/bench.py --beam 4 --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-l8 > $GRAFT_REPO_ROOT/gpurun_out/r2n_beam4_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2n_rocprof.err)
find gpurun_out/prof_r2n -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r2n -name "*.db" -delete
f=$(find gpurun_out/prof_r2n -name "*kernel_stats.csv" | head -1); head -22 "$f" | cut -c1-230
cut -c1-200 gpurun_out/r2n_beam4_under_rocprof.json
