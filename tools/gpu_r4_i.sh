cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "streaming or prologue or stream" > $O/i_pytest_ops.log 2>&1; tail -8 $O/i_pytest_ops.log
timeout 600 python -m pytest tests/test_stream_engine_gpu.py tests/test_generator_gpu.py -q -x > $O/i_pytest_engine.log 2>&1; tail -8 $O/i_pytest_engine.log
{
for v in 0 3 1 2 0 3; do
echo "== CT2X_STREAM_PROLOGUE=$v"; CT2X_STREAM_PROLOGUE=$v timeout 400 python tools/bench_l8.py --steps 3 --warmup 1 --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],1),'tok/s decode_ms_per_step',round(c['decode_ms_per_step'],4),'prefill_ms',round(c['prefill_ms_last_call'],2),'kernels/step',c['kernels_per_decode_step'],'load_s',round(c['model_load_s'],1),'build_s',round(c['model_build_s'],1))"
done
} > $O/i_l8_ab.txt 2>&1
cat $O/i_l8_ab.txt
