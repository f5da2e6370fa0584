cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "streaming or prologue or stream" > $O/j_pytest_ops.log 2>&1; tail -3 $O/j_pytest_ops.log
export TMPDIR=/tmp
for v in 0 3; do
( cd /tmp && CT2X_STREAM_PROLOGUE=$v timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $GRAFT_REPO_ROOT/tools/bench_l8.py --steps 1 --warmup 1 --no-profile > /tmp/prof_$v.log 2>&1 )
f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
cp "$f" $O/j_kernel_stats_l8_prologue_$v.csv
tail -1 /tmp/prof_$v.log | cut -c1-300
head -12 $O/j_kernel_stats_l8_prologue_$v.csv | cut -c1-200
done
