cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/chain_probe.cpp -o /tmp/cp 2>&1 | tail -3
/tmp/cp > gpurun_out/r2f_chain_probe.txt 2>&1; cat gpurun_out/r2f_chain_probe.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>&1 | tail -3
/tmp/kb 32 2>&1 | grep -E "softmax|topk"
timeout 900 python -m pytest tests/test_translator_gpu.py -m gpu -q -s -k "free_running" 2>&1 | grep -E "identical|passed|failed" | cut -c1-200
