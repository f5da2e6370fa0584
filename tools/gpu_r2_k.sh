cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "norm_gemm" > gpurun_out/r2k_ops.log 2>&1; tail -4 gpurun_out/r2k_ops.log
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null && /tmp/kb 32 > gpurun_out/r2k_kernel_bench.txt 2>&1; grep -i "norm\|gemm_s8_dequant \[M=32,N=512,K=512\]" gpurun_out/r2k_kernel_bench.txt
python bench.py --no-cpu-baseline --no-l8 > gpurun_out/r2k_bench.json 2> gpurun_out/r2k_bench.err; cut -c1-300 gpurun_out/r2k_bench.json; tail -2 gpurun_out/r2k_bench.err
timeout 900 python -m pytest tests/test_translator_gpu.py tests/test_parity_taps_gpu.py -m gpu -q > gpurun_out/r2k_parity.log 2>&1; tail -4 gpurun_out/r2k_parity.log
