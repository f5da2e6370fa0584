cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r2f_pytest.log 2>&1; tail -4 gpurun_out/r2f_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r2f_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('kernels_per_decode_step')); print(json.dumps(d['roofline'])[:600]); print(json.dumps(d['config'].get('l8'))[:500]); print(json.dumps(d.get('cpu_baseline'))[:400]); print(json.dumps(d['config'].get('beam4'))[:200])"
mkdir -p gpurun_out/prof_r2f
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2f -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-l8 > $GRAFT_REPO_ROOT/gpurun_out/r2f_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2f_rocprof.err)
find gpurun_out/prof_r2f -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r2f -name "*.db" -delete
f=$(find gpurun_out/prof_r2f -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-90,150-230
