cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -x hip tools/gemm_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/gb 2>/dev/null
SH="16 14336 4096 16 4096 14336 16 6144 4096 16 4096 4096"
(echo "== weights rotated through 600 MB (HBM)"; /tmp/gb $SH | grep dequant; echo "== same weights every launch (MALL resident)"; GB_NO_ROTATE=1 /tmp/gb $SH | grep dequant;
 echo "== CT2X_STREAM_GEMM=0 rotated"; CT2X_STREAM_GEMM=0 /tmp/gb $SH | grep dequant; echo "== CT2X_STREAM_GEMM=0 MALL resident"; CT2X_STREAM_GEMM=0 GB_NO_ROTATE=1 /tmp/gb $SH | grep dequant) > gpurun_out/r2ar_mall.txt 2>&1
cat gpurun_out/r2ar_mall.txt
