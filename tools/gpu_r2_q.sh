cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null
for mt in 1 2; do echo "== M=32 CT2X_SKINNY_MT=$mt"; CT2X_SKINNY_MT=$mt /tmp/kb 32 2>&1 | grep "gemm_s8_dequant"; done
for mt in 1 2; do
  echo "== beam1 MT=$mt"; CT2X_SKINNY_MT=$mt timeout 300 python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | cut -c1-190
done
for mt in 1 2; do
  echo "== beam1 MT=$mt"; CT2X_SKINNY_MT=$mt timeout 300 python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | cut -c1-190
done
