cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x > gpurun_out/r2o_ops.log 2>&1; tail -4 gpurun_out/r2o_ops.log
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null && /tmp/kb 32 > gpurun_out/r2o_kernel_bench.txt 2>&1; grep -i "norm\|quantize\|gemm_s8_dequant \[M=32,N=512,K=512\]" gpurun_out/r2o_kernel_bench.txt
/tmp/kb 128 > gpurun_out/r2o_kernel_bench_m128.txt 2>&1; grep -i "norm\|quantize\|gemm_s8_dequant \[M=128,N=512,K=512\]" gpurun_out/r2o_kernel_bench_m128.txt
for i in 1 2; do
for f in 1 0; do
  echo "== NORM_FUSION=$f"; CT2X_NORM_FUSION=$f timeout 300 python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | cut -c1-190
done
done
for f in 1 0; do
  echo "== beam4 NORM_FUSION=$f"; CT2X_NORM_FUSION=$f timeout 300 python bench.py --beam 4 --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | cut -c1-190
done
timeout 900 python -m pytest tests/test_translator_gpu.py tests/test_parity_taps_gpu.py tests/test_reference_kats_gpu.py -m gpu -q > gpurun_out/r2o_parity.log 2>&1; tail -4 gpurun_out/r2o_parity.log
python tools/bench_l8.py --steps 2 --warmup 1 --no-profile 2>/dev/null | cut -c1-600
