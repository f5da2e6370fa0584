cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
GB=build/tools/gemm_bench
SH="4096 2048 512 4096 1536 512 4096 512 2048 8192 2048 512 12000 1280 1280 12000 5120 1280 12000 1280 5120 12000 3840 1280 2048 4096 1024 2048 14336 1024 4096 4096 2048"
{
echo "== v4 (default)"; timeout 300 $GB $SH
echo "== v5"; CT2X_BIG_GEMM_V=5 timeout 300 $GB $SH
echo "== v1"; CT2X_BIG_GEMM_V=1 timeout 300 $GB $SH
} > $O/h_gemm_bench.txt 2>&1
cat $O/h_gemm_bench.txt
