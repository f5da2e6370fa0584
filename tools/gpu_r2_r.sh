cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -x hip tools/attn_probe.cpp -Ictranslate2_amd/csrc -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/ap 2>/dev/null && /tmp/ap | tee gpurun_out/r2r_attn_probe.txt
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null
for cfg in 1,4,4 1,4,8 1,8,4 1,8,8 1,16,4 2,4,4 2,8,4; do echo "== M=32 CFG=$cfg"; CT2X_SKINNY_CFG=$cfg /tmp/kb 32 2>&1 | grep "gemm_s8_dequant \[M=32,N=512\|gemm_s8_dequant \[M=32,N=2048\|rotating"; done
