#!/bin/bash
# Round 5: the GPU measurements of the round as named sections (one `gpurun` call each):
#   gpurun --timeout 900 -- 'bash tools/gpu_r5.sh <section>'
# Output goes to gpurun_out/r5/; what is kept is copied to profiles/ (profiles/README.md).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
O=gpurun_out/r5
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
NOLEGS="--no-cpu-baseline --no-profile --no-l8 --no-whisper --no-nllb"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/call', 'kernels/step', d['config'].get('kernels_per_decode_step'))"; }

case "$1" in
probes)           # profiles/r05_row_resident_probe.txt, r05_persistent_layer_probe.txt
  timeout 120 build/tools/row_resident_probe 2>&1 | tee $O/row_resident_probe.txt
  timeout 120 build/tools/persistent_layer_probe 2>&1 | tee $O/persistent_layer_probe.txt ;;
cluster_first)    # first run of the cluster-resident decoder stack: parity, then the A/B
  timeout 600 python -m pytest tests/test_decoder_cluster_gpu.py tests/test_bench_path_gpu.py -x -q 2>&1 | tail -15 | tee $O/cluster_first_tests.txt
  { for f in 0 1 0 1; do echo "== CT2X_CLUSTER_STACK=$f"; CT2X_CLUSTER_STACK=$f timeout 300 python bench.py --steps 20 --warmup 3 $NOLEGS 2>&1 | tail -1 | line; done
  } 2>&1 | tee $O/cluster_ab.txt ;;
cluster_iter)     # parity of the cluster kernel, its phase timeline, the A/B
  timeout 600 python -m pytest tests/test_decoder_cluster_gpu.py tests/test_bench_path_gpu.py -x -q 2>&1 | tail -5 | tee $O/cluster_iter_tests.txt
  CT2X_CLUSTER_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 1 $NOLEGS --no-token-check 2>&1 | grep -A8 "decoder_stack timeline" | tail -9 | tee $O/cluster_timeline.txt
  { for f in 0 1 1; do echo "== CT2X_CLUSTER_STACK=$f"; CT2X_CLUSTER_STACK=$f timeout 300 python bench.py --steps 20 --warmup 3 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
    for h in counter tagged counter tagged; do echo "== CT2X_CLUSTER_HANDOVER=$h"; CT2X_CLUSTER_HANDOVER=$h timeout 300 python bench.py --steps 20 --warmup 3 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
  } 2>&1 | tee $O/cluster_ab.txt ;;
cluster_ab)       # profiles/r05_cluster_stack_ab.txt
  { for f in 0 1 0 1; do echo "== CT2X_CLUSTER_STACK=$f $2"; CT2X_CLUSTER_STACK=$f timeout 300 python bench.py --steps 20 --warmup 3 $NOLEGS $2 2>&1 | tail -1 | line; done
  } 2>&1 | tee $O/cluster_ab.txt ;;
tests)            # a list of test files
  shift; timeout 1200 python -m pytest "$@" -x -q 2>&1 | tail -15 | tee $O/tests.txt ;;
rocprof_tb)       # profiles/r05_kernel_stats_bench_tb_b32.csv
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tb -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 $NOLEGS --no-token-check > /tmp/prof_tb.log 2>&1 )
  cp "$(find /tmp/prof_tb -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_bench_tb_b32.csv; head -12 $O/kernel_stats_bench_tb_b32.csv | cut -c1-200 ;;
beam_iter)        # the beam-search cluster launch: parity, then the A/B of bench.py --beam 4
  timeout 900 python -m pytest tests/test_decoder_cluster_gpu.py -x -q 2>&1 | tail -12 | tee $O/beam_iter_tests.txt
  CT2X_CLUSTER_STAMPS=1 timeout 300 python bench.py --beam 4 --steps 2 --warmup 1 $NOLEGS --no-token-check 2>&1 | grep -A8 "decoder_stack timeline" | tail -9 | tee $O/beam_timeline.txt
  { for f in 0 1; do echo "== CT2X_CLUSTER_BEAM=$f --beam 4"; CT2X_CLUSTER_BEAM=$f timeout 300 python bench.py --beam 4 --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
  } 2>&1 | tee $O/beam_ab.txt ;;
many_rows)        # greedy batches of 64 / 128 rows: several rows per cluster against the launch-per-op step
  { for b in 64 96 128; do for f in 0 1; do echo "== batch $b CT2X_CLUSTER_STACK=$f (CT2X_CLUSTER_MAX_ROWS=128)"; CT2X_CLUSTER_MAX_ROWS=128 CT2X_CLUSTER_STACK=$f timeout 300 python bench.py --batch $b --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done; done
  } 2>&1 | tee $O/many_rows_ab.txt ;;
evidence)         # profiles/r05_tb_kernel_trace_gaps.txt, r05_pmc_stream_gemm.json: the round's counter / gap refresh
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gaps -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 $NOLEGS --no-token-check > /tmp/prof_gaps.log 2>&1 )
  f=$(ls -S $(find /tmp/prof_gaps -name '*kernel_trace.csv') | head -1)
  python tools/trace_gaps.py "$f" 2>&1 | head -40 | tee $O/tb_kernel_trace_gaps.txt
  timeout 600 python tools/pmc_stream_gemm.py $O/pmc_stream > $O/pmc_stream_gemm.json 2> $O/pmc_stream.err; head -c 1500 $O/pmc_stream_gemm.json; find $O/pmc_stream -name "*.csv" -size +1M -delete; find $O/pmc_stream -name "*.db" -delete ;;
pmc)              # profiles/r05_pmc_traffic.json
  timeout 600 python tools/pmc_bench.py $O/pmc > $O/pmc_traffic.json; cat $O/pmc_traffic.json | head -30; find $O/pmc -name "*.csv" -size +1M -delete; find $O/pmc -name "*.db" -delete ;;
level15)          # Level 1.5: the reference's own Translator on the fused engine
  timeout 900 python -m pytest tests/test_level1_gpu.py -x -q 2>&1 | tail -8 | tee $O/level15_tests.txt
  { for f in 0 1; do echo "== CT2X_LEVEL1_FUSED=$f"; CT2X_LEVEL1_FUSED=$f timeout 300 python tools/level1_speed.py 2>&1 | tail -1; done
  } | tee $O/level15_speed.txt ;;
suite)            # the whole -m gpu suite only
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; tail -5 $O/pytest_all.log ;;
final_prof)       # profiles/r05_final_kernel_stats_*: rocprofv3 kernel stats of the TB legs
  for leg in "b32" "beam4 --beam 4"; do
    set -- $leg; name=$1; shift
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 $NOLEGS --no-token-check "$@" > /tmp/prof_$name.log 2>&1 )
    f=$(ls -S $(find /tmp/prof_$name -name '*kernel_stats.csv') | head -1)
    cp "$f" $O/final_kernel_stats_bench_tb_$name.csv; echo "== $name"; head -8 $O/final_kernel_stats_bench_tb_$name.csv | cut -c1-170
  done ;;
attn_h)           # sequence-mode attention on the 16-bit matrix cores: parity, the 16-bit model tests, the Whisper line + its kernel stats
  timeout 900 python -m pytest tests/test_ops_gpu.py -k "attention_full or cross_attention or gemm_half" -x -q 2>&1 | tail -5 | tee $O/attn_h_tests.txt
  timeout 1200 python -m pytest tests/test_whisper_gpu.py tests/test_translator_gpu.py tests/test_generator_gpu.py -k "half or float16 or bfloat16 or bf16 or fp16 or whisper or nllb or 16" -x -q 2>&1 | tail -5 | tee -a $O/attn_h_tests.txt
  timeout 600 python tools/bench_whisper.py 2>&1 | tail -1 | tee $O/whisper_attn_h.json
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wh -- python $GRAFT_REPO_ROOT/tools/bench_whisper.py --steps 1 --warmup 1 > /tmp/prof_wh.log 2>&1 )
  f=$(ls -S $(find /tmp/prof_wh -name '*kernel_stats.csv') | head -1); cp "$f" $O/kernel_stats_bench_whisper.csv; head -10 $O/kernel_stats_bench_whisper.csv | cut -c1-200 ;;
xattn_probe)      # profiles/r05_xattn_probe.txt: the 16-bit cross attention at the Whisper shape, chunks 1 .. 8 (TUNING library)
  CT2X_LIBRARY=$GRAFT_REPO_ROOT/ctranslate2_amd/libct2_mi355_tuning.so timeout 300 python tools/xattn_probe.py 2>&1 | tee $O/xattn_probe.txt ;;
rows_probe)       # profiles/r05_gemm_rows_probe.txt: the decode-row 16-bit GEMM at the Whisper shapes (TUNING library)
  CT2X_LIBRARY=$GRAFT_REPO_ROOT/ctranslate2_amd/libct2_mi355_tuning.so timeout 300 python tools/gemm_rows_probe.py 2>&1 | tee $O/gemm_rows_probe.txt ;;
final_legs)       # profiles/r05_final_kernel_stats_bench_{nllb,whisper}.csv: rocprofv3 kernel stats of the 16-bit legs' own processes
  for leg in nllb whisper; do
    ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$leg -- python $GRAFT_REPO_ROOT/tools/bench_$leg.py --steps 1 --warmup 1 > /tmp/prof_$leg.log 2>&1 )
    f=$(ls -S $(find /tmp/prof_$leg -name '*kernel_stats.csv') | head -1)
    cp "$f" $O/final_kernel_stats_bench_$leg.csv; echo "== $leg"; tail -1 /tmp/prof_$leg.log | cut -c1-260; head -7 $O/final_kernel_stats_bench_$leg.csv | cut -c1-150
  done ;;
pmc_whisper)      # profiles/r05_pmc_whisper.json
  timeout 700 python tools/pmc_whisper.py $O/pmc_whisper > $O/pmc_whisper.json; cat $O/pmc_whisper.json | head -40; find $O/pmc_whisper -name "*.csv" -size +1M -delete; find $O/pmc_whisper -name "*.db" -delete ;;
all)              # the whole -m gpu suite, smoke and the default bench line
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; tail -3 $O/pytest_all.log
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json ;;
*) echo "sections: probes cluster_first cluster_iter cluster_ab tests rocprof_tb beam_iter many_rows evidence pmc level15 suite final_prof attn_h xattn_probe rows_probe final_legs pmc_whisper all"; exit 2 ;;
esac
