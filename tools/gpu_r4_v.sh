cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
{
for i in 1 2 3; do
echo "== previous library"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libct2_prev.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile --no-l8 --no-whisper --no-nllb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/call')"
echo "== pinned staging buffers cached"; timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile --no-l8 --no-whisper --no-nllb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/call')"
done
for lib in $GRAFT_REPO_ROOT/build_ab/libct2_prev.so ""; do
echo "== beam 4, library: ${lib:-this build}"; CT2X_LIBRARY=$lib timeout 300 python bench.py --beam 4 --steps 20 --warmup 3 --no-cpu-baseline --no-profile --no-l8 --no-whisper --no-nllb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/call')"
done
} > $O/v_pinned_cache_ab.txt 2>&1
cat $O/v_pinned_cache_ab.txt
