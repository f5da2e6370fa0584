cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm_half_weights" > $O/k_pytest_gemm_h.log 2>&1; tail -6 $O/k_pytest_gemm_h.log
timeout 600 python -m pytest tests/test_translator_gpu.py -q -x -k "nllb or float16 or half or bfloat16" > $O/k_pytest_tr.log 2>&1; tail -4 $O/k_pytest_tr.log
timeout 900 python tools/bench_nllb.py --steps 2 --warmup 1 > $O/k_nllb.json 2> $O/k_nllb.err; tail -c 3000 $O/k_nllb.json
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_n -- python $GRAFT_REPO_ROOT/tools/bench_nllb.py --steps 1 --warmup 1 --no-profile > /tmp/prof_n.log 2>&1 )
f=$(find /tmp/prof_n -name "*kernel_stats.csv" | head -1); cp "$f" $O/k_kernel_stats_nllb.csv; head -12 $O/k_kernel_stats_nllb.csv | cut -c1-180
