cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_aq_l8 -- python $R/tools/bench_l8.py --steps 1 --warmup 1 --no-profile > $R/gpurun_out/r2aq_l8.json 2> $R/gpurun_out/r2aq_l8.err)
f=$(find gpurun_out/prof_aq_l8 -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py "$f" > gpurun_out/r2aq_l8_gaps.txt 2>&1
find gpurun_out/prof_aq_l8 -name "*.csv" -delete; find gpurun_out/prof_aq_l8 -name "*.db" -delete
head -12 gpurun_out/r2aq_l8_gaps.txt
