cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
GB=build/tools/gemm_bench
SH="2048 14336 4096 4096 4096 4096 8192 8192 8192 2048 4096 14336"
{
echo "== v5 (two workgroups per CU)"; CT2X_BIG_GEMM_V=5 timeout 300 $GB $SH
echo "== v5 K loops only"; CT2X_BIG_GEMM_V=5 CT2X_BIG5_VAR=2 timeout 300 $GB $SH
echo "== v4 persistent (default)"; timeout 300 $GB $SH
echo "== v1 (round 3)"; CT2X_BIG_GEMM_V=1 timeout 300 $GB $SH
echo "== v5, swish"; CT2X_BIG_GEMM_V=5 timeout 300 $GB --act 2 2048 14336 4096 4096 4096 4096
echo "== v4, swish"; timeout 300 $GB --act 2 2048 14336 4096 4096 4096 4096
echo "== v5 edges"; CT2X_BIG_GEMM_V=5 timeout 300 $GB 2500 2604 640 2500 2601 384 4200 4404 384 2561 2817 256
echo "== v5 edges no act"; CT2X_BIG_GEMM_V=5 timeout 300 $GB --act -1 2500 2604 640 2304 2816 512
} > $O/d_gemm_bench.txt 2>&1
cat $O/d_gemm_bench.txt
