cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "attention or softmax" 2>&1 | tail -3
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -x hip tools/attn_probe.cpp -Ictranslate2_amd/csrc -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/ap 2>/dev/null && /tmp/ap | tee gpurun_out/r2s_attn_probe.txt
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null && /tmp/kb 32 > gpurun_out/r2s_kernel_bench.txt; grep attention gpurun_out/r2s_kernel_bench.txt
for i in 1 2; do python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | cut -c1-190; done
python bench.py --beam 4 --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | cut -c1-190
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2s_pytest.log 2>&1; tail -5 gpurun_out/r2s_pytest.log
