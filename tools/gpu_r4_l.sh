cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_n -- python $GRAFT_REPO_ROOT/tools/bench_nllb.py --steps 1 --warmup 1 > /tmp/prof_n.log 2>&1 ); tail -3 /tmp/prof_n.log | cut -c1-300
f=$(find /tmp/prof_n -name "*kernel_stats.csv" | head -1); cp "$f" $O/l_kernel_stats_nllb.csv; head -14 $O/l_kernel_stats_nllb.csv | cut -c1-220
