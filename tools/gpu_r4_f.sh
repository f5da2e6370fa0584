cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
GB=build/tools/gemm_bench
SH="2048 14336 4096 4096 4096 4096 8192 8192 8192 2048 4096 14336 2048 6144 4096"
{
echo "== v4 stream-K (default)"; timeout 300 $GB $SH
echo "== v4 stream-K off"; CT2X_BIG3_SK=0 timeout 300 $GB $SH
echo "== v4 stream-K, K loops only"; CT2X_BIG3_VAR=2 timeout 300 $GB 2048 14336 4096
echo "== v4 stream-K with every slab treated as foreign (recompute arm)"; CT2X_BIG3_SK=3 timeout 300 $GB 2048 14336 4096 4200 4404 384 5000 5000 512
echo "== v1 (round 3)"; CT2X_BIG_GEMM_V=1 timeout 300 $GB 2048 14336 4096 4096 4096 4096
echo "== v4 stream-K, swish"; timeout 300 $GB --act 2 2048 14336 4096
echo "== v4 stream-K, no act"; timeout 300 $GB --act -1 2048 14336 4096
echo "== v4 edges"; timeout 300 $GB 2500 2604 640 2500 2601 384 4200 4404 384 2561 2817 256 5000 5000 512
echo "== v4 grid 200 / 100 (uneven)"; CT2X_BIG3_GRID=200 timeout 300 $GB 2048 14336 4096 2500 2604 640; CT2X_BIG3_GRID=100 timeout 300 $GB 2048 14336 4096 2500 2604 640 4200 4404 384
} > $O/f_gemm_bench.txt 2>&1
cat $O/f_gemm_bench.txt
for i in 1 2 3; do timeout 120 $GB 2048 14336 4096 4200 4404 384 5000 5000 512 | grep -c "mismatches=0/"; done
