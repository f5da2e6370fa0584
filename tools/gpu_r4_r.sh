cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
{
for v in 0 64 0 64; do CT2X_COMPACT_MIN_ROWS=$v timeout 600 python tools/bench_compaction.py 2>/dev/null | tail -1; done
for v in 0 64; do BENCH_ROWS=64 CT2X_COMPACT_MIN_ROWS=$v timeout 600 python tools/bench_compaction.py 2>/dev/null | tail -1; done
} > $O/r_compaction_bench.txt 2>&1
cat $O/r_compaction_bench.txt
