cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -x hip tools/gemm_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/gb 2>/dev/null
SH="4096 4096 4096 2048 14336 4096 8192 8192 8192"
(for ab in 0 5 4 1 3 2; do echo "== CT2X_BIG_GEMM_ABLATE=$ab"; CT2X_BIG_GEMM_ABLATE=$ab /tmp/gb $SH | sed 's/mismatches.*//'; done) > gpurun_out/r2ak_gemm_ablate.txt 2>&1
cat gpurun_out/r2ak_gemm_ablate.txt
