cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "norm_gemm or runtime_abi" > gpurun_out/r2j_ops.log 2>&1; tail -4 gpurun_out/r2j_ops.log
timeout 900 python -m pytest tests/test_translator_gpu.py tests/test_parity_taps_gpu.py tests/test_reference_kats_gpu.py -m gpu -q > gpurun_out/r2j_parity.log 2>&1; tail -4 gpurun_out/r2j_parity.log
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null && /tmp/kb 32 > gpurun_out/r2j_kernel_bench.txt 2>&1; grep -i "norm\|gemm_s8_dequant \[M=32,N=512,K=512\]" gpurun_out/r2j_kernel_bench.txt
python bench.py --no-cpu-baseline --no-l8 > gpurun_out/r2j_bench.json 2> gpurun_out/r2j_bench.err; cut -c1-400 gpurun_out/r2j_bench.json; tail -2 gpurun_out/r2j_bench.err
CT2X_NORM_FUSION=0 python bench.py --no-cpu-baseline --no-l8 --no-profile > gpurun_out/r2j_bench_nofuse.json 2>/dev/null; cut -c1-200 gpurun_out/r2j_bench_nofuse.json
