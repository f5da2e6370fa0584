cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>&1 | tail -3
/tmp/kb 32 > gpurun_out/r2g_kernel_bench.txt 2>&1; grep -E "norm|attention|softmax|quantize" gpurun_out/r2g_kernel_bench.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2g_pytest.log 2>&1; tail -8 gpurun_out/r2g_pytest.log
python bench.py --no-cpu-baseline > gpurun_out/r2g_bench.json 2> gpurun_out/r2g_bench.err; cut -c1-260 gpurun_out/r2g_bench.json; python -c "
import json; d=json.loads(open('gpurun_out/r2g_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d['config'].get('beam4'))[:300]); print(json.dumps(d.get('profile'))[:1800])"
