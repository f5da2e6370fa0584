cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null && /tmp/kb 32 > gpurun_out/r2at_kernel_bench.txt; grep gemm gpurun_out/r2at_kernel_bench.txt
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('kernels_per_decode_step'), (d['config'].get('beam4') or {}).get('value'))"; done
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2at_pytest.log 2>&1; tail -3 gpurun_out/r2at_pytest.log
