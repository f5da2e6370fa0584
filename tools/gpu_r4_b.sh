cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
GB=build/tools/gemm_bench
SH="2048 14336 4096 2048 4096 4096 2048 4096 14336 4096 4096 4096 8192 8192 8192"
{
echo "== v4 persistent (default), uniform fill"; timeout 300 $GB $SH
echo "== v4 one workgroup per tile, uniform fill"; CT2X_BIG3_GRID=0 timeout 300 $GB $SH
echo "== v1 (round 3), uniform fill"; CT2X_BIG_GEMM_V=1 timeout 300 $GB $SH
echo "== v4 persistent, swish, uniform"; timeout 300 $GB --act 2 2048 14336 4096 4096 4096 4096
echo "== v1, swish, uniform"; CT2X_BIG_GEMM_V=1 timeout 300 $GB --act 2 2048 14336 4096 4096 4096 4096
echo "== v4 edges"; timeout 300 $GB 2500 2604 640 2500 2601 384 2561 2817 256 3000 2560 1024
echo "== v4 edges, no activation"; timeout 300 $GB --act -1 2500 2604 640 2304 2816 512
echo "== v4 grid 200 (uneven rounds)"; CT2X_BIG3_GRID=200 timeout 300 $GB 2048 14336 4096 2500 2604 640
} > $O/b_gemm_bench.txt 2>&1
cat $O/b_gemm_bench.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm or dense" > $O/b_pytest_gemm.log 2>&1; tail -3 $O/b_pytest_gemm.log
