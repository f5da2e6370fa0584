cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null
for mt in 1 2 4 8; do echo "== M=128 CT2X_SKINNY_MT=$mt"; CT2X_SKINNY_MT=$mt /tmp/kb 128 2>&1 | grep "gemm_s8_dequant \[M=128"; done
for mt in 1 2 4; do echo "== M=64 CT2X_SKINNY_MT=$mt"; CT2X_SKINNY_MT=$mt /tmp/kb 64 2>&1 | grep "gemm_s8_dequant \[M=64"; done
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm" 2>&1 | tail -2
for mt in 1 2 8; do
  echo "== beam4 MT=$mt"; CT2X_SKINNY_MT=$mt timeout 300 python bench.py --beam 4 --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | cut -c1-190
done
