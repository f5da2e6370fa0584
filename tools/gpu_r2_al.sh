cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_al_l8 -- python $R/tools/bench_l8.py --steps 1 --warmup 1 --no-profile > $R/gpurun_out/r2al_l8.json 2> $R/gpurun_out/r2al_l8.err)
f=$(find gpurun_out/prof_al_l8 -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py "$f" > gpurun_out/r2al_l8_gaps.txt 2>&1; head -3 "$f" > gpurun_out/r2al_trace_head.txt
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_al_tb -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-l8 > $R/gpurun_out/r2al_tb.json 2> $R/gpurun_out/r2al_tb.err)
f=$(find gpurun_out/prof_al_tb -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py "$f" > gpurun_out/r2al_tb_gaps.txt 2>&1
find gpurun_out/prof_al_l8 gpurun_out/prof_al_tb -name "*.csv" -delete; find gpurun_out/prof_al_l8 gpurun_out/prof_al_tb -name "*.db" -delete
cat gpurun_out/r2al_l8_gaps.txt; cat gpurun_out/r2al_tb_gaps.txt; cat gpurun_out/r2al_trace_head.txt | cut -c1-600
