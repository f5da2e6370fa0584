# Round 4, GPU call A: matrix-instruction rates by operand data; the round-4 big GEMM (v3) against
# the round-3 kernel (v1) in a tuning build; GEMM op tests with the production library.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
timeout 120 build/tools/mfma_rate_probe 4096 > $O/a_mfma_rate_probe.txt 2>&1; cat $O/a_mfma_rate_probe.txt
GB=build/tools/gemm_bench
SH="2048 14336 4096 2048 4096 4096 2048 4096 14336 4096 4096 4096 8192 8192 8192 1024 1024 512"
{
echo "== v3 persistent (default), uniform fill"; timeout 300 $GB $SH
echo "== v3 one workgroup per tile, uniform fill"; CT2X_BIG3_GRID=0 timeout 300 $GB $SH
echo "== v1 (round 3), uniform fill"; CT2X_BIG_GEMM_V=1 timeout 300 $GB $SH
echo "== v3 persistent, gauss fill"; timeout 300 $GB --fill gauss $SH
echo "== v1, gauss fill"; CT2X_BIG_GEMM_V=1 timeout 300 $GB --fill gauss $SH
echo "== v3 persistent, swish, uniform"; timeout 300 $GB --act 2 2048 14336 4096 4096 4096 4096
echo "== v1, swish, uniform"; CT2X_BIG_GEMM_V=1 timeout 300 $GB --act 2 2048 14336 4096 4096 4096 4096
echo "== v3 persistent, swish, gauss"; timeout 300 $GB --fill gauss --act 2 2048 14336 4096 4096 4096 4096
echo "== v3 edges"; timeout 300 $GB 2500 2604 640 2500 2601 384 2561 2817 256 3000 2560 1024
echo "== v3 edges, no activation"; timeout 300 $GB --act -1 2500 2604 640 2304 2816 512
echo "== v3 grid 200 (uneven rounds)"; CT2X_BIG3_GRID=200 timeout 300 $GB 2048 14336 4096 2500 2604 640
} > $O/a_gemm_bench.txt 2>&1
cat $O/a_gemm_bench.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm or dense" > $O/a_pytest_gemm.log 2>&1; tail -5 $O/a_pytest_gemm.log
