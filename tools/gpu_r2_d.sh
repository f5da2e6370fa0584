cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>&1 | tail -3
/tmp/kb 32 > gpurun_out/r2d_kernel_bench.txt 2>&1; cat gpurun_out/r2d_kernel_bench.txt
CT2X_FAST_ATTENTION=1 /tmp/kb 32 2>&1 | grep attention > gpurun_out/r2d_kernel_bench_fastattn.txt; cat gpurun_out/r2d_kernel_bench_fastattn.txt
timeout 900 python -m pytest tests/test_translator_gpu.py -m gpu -q -s -k "free_running or teacher_forced_logits_int8" 2>&1 | grep -E "identical|flip|passed|failed" | cut -c1-200
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k softmax 2>&1 | tail -2
