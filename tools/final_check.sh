cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_final.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_final.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cut -c1-200 gpurun_out/bench_final.json
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_final
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_final -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile > $GRAFT_REPO_ROOT/gpurun_out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/rocprof.err)
find gpurun_out/prof_final -name "*kernel_stats.csv" | head -2
find gpurun_out/prof_final -name "*kernel_trace.csv" -delete
find gpurun_out/prof_final -name "*.db" -delete
du -sh gpurun_out/prof_final
