cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r2z_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r2z_pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r2z_bench.json 2> gpurun_out/r2z_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r2z_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('kernels_per_decode_step')); print(json.dumps(d['roofline'])[:700]); print(json.dumps(d['config'].get('l8'))[:600]); print(json.dumps(d.get('cpu_baseline'))[:300]); print(json.dumps(d['config'].get('beam4'))[:200])"
mkdir -p gpurun_out/prof_r2z
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2z -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-l8 > $R/gpurun_out/r2z_bench_under_rocprof.json 2> $R/gpurun_out/r2z_rocprof.err)
f=$(find gpurun_out/prof_r2z -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py "$f" > gpurun_out/r2z_tb_gaps.txt 2>&1
find gpurun_out/prof_r2z -name "*kernel_trace.csv" -delete; find gpurun_out/prof_r2z -name "*.db" -delete
f=$(find gpurun_out/prof_r2z -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2z_kernel_stats.csv; head -8 "$f" | cut -c1-100,150-230
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null && /tmp/kb 32 > gpurun_out/r2z_kernel_bench.txt
timeout 300 python tools/pmc_traffic.py gpurun_out/pmc_r2z > gpurun_out/r2z_pmc_traffic.json 2> gpurun_out/r2z_pmc.err; head -c 600 gpurun_out/r2z_pmc_traffic.json
find gpurun_out/pmc_r2z -name "*.csv" -size +2M -delete; find gpurun_out/pmc_r2z -name "*.db" -delete
