cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_translator_gpu.py tests/test_generator_gpu.py tests/test_reference_kats_gpu.py -q -x > gpurun_out/r2x_parity.log 2>&1; tail -3 gpurun_out/r2x_parity.log
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "softmax or beam or greedy" 2>&1 | tail -2
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null && /tmp/kb 32 | grep -i "softmax\|topk"
for i in 1 2; do python bench.py --beam 4 --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | cut -c1-190; done
python bench.py --no-cpu-baseline --no-l8 --no-profile --streams 2 2>/dev/null | cut -c1-190
