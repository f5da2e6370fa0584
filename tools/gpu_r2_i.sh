cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r2i_pytest.log 2>&1; tail -4 gpurun_out/r2i_pytest.log
python bench.py > gpurun_out/r2i_bench.json 2> gpurun_out/r2i_bench.err; cut -c1-300 gpurun_out/r2i_bench.json; tail -3 gpurun_out/r2i_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r2i_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['config'].get('l8'))[:700]); print(json.dumps(d.get('cpu_baseline'))[:900]); print(json.dumps(d['config'].get('beam4'))[:200])"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r2i
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2i -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-l8 > $GRAFT_REPO_ROOT/gpurun_out/r2i_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2i_rocprof.err)
find gpurun_out/prof_r2i -name "*kernel_stats.csv" | head -2
find gpurun_out/prof_r2i -name "*kernel_trace.csv" -delete
find gpurun_out/prof_r2i -name "*.db" -delete
f=$(find gpurun_out/prof_r2i -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-200
