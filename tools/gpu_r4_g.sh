cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
GB=build/tools/gemm_bench
{
echo "== v4 (default)"; timeout 300 $GB 2048 14336 4096 4096 4096 4096 8192 8192 8192 2048 4096 14336 2048 6144 4096 2048 4096 4096 1024 2048 512
echo "== v1 (round 3)"; CT2X_BIG_GEMM_V=1 timeout 300 $GB 2048 14336 4096 4096 4096 4096 8192 8192 8192 2048 4096 14336 2048 6144 4096 2048 4096 4096
echo "== v4, swish"; timeout 300 $GB --act 2 2048 14336 4096
echo "== v1, swish"; CT2X_BIG_GEMM_V=1 timeout 300 $GB --act 2 2048 14336 4096
echo "== v4 edges"; timeout 300 $GB 2500 2604 640 2500 2601 384 4200 4404 384 2561 2817 256 5000 5000 512
} > $O/g_gemm_bench.txt 2>&1
cat $O/g_gemm_bench.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/g_pytest_full.log 2>&1; tail -4 $O/g_pytest_full.log
