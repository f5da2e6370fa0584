cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
GB=build/tools/gemm_bench
SH="2048 14336 4096 4096 4096 4096 8192 8192 8192"
{
echo "== v4 persistent (default)"; timeout 300 $GB $SH
echo "== v4 one workgroup per tile"; CT2X_BIG3_GRID=0 timeout 300 $GB $SH
echo "== v4 VAR=1 (weight fragment outer)"; CT2X_BIG3_VAR=1 timeout 300 $GB $SH
echo "== v4 VAR=2 (K loops only)"; CT2X_BIG3_VAR=2 timeout 300 $GB $SH
echo "== v4 VAR=3 (K loops only, weight fragment outer)"; CT2X_BIG3_VAR=3 timeout 300 $GB $SH
echo "== v1 (round 3)"; CT2X_BIG_GEMM_V=1 timeout 300 $GB $SH
echo "== v1 K loop only"; CT2X_BIG_GEMM_V=1 CT2X_BIG_GEMM_ABLATE=5 timeout 300 $GB $SH
echo "== v4 persistent, swish"; timeout 300 $GB --act 2 2048 14336 4096
echo "== v4 edges"; timeout 300 $GB 2500 2604 640 2500 2601 384 4200 4404 384
} > $O/c_gemm_bench.txt 2>&1
cat $O/c_gemm_bench.txt
