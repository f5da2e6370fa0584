#!/bin/bash
# Round 4: every GPU measurement of the round as a named section (one `gpurun` call each):
#   gpurun --timeout 1800 -- 'bash tools/gpu_r4.sh <section>'
# Output goes to gpurun_out/r4/; what was kept is copied to profiles/ (profiles/README.md names the
# section that produced each file). The tuning sections need `make TUNING=1` libraries / the tools
# under build/tools (tools/*.cpp).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
GB=build/tools/gemm_bench
NOLEGS="--no-cpu-baseline --no-profile --no-l8 --no-whisper --no-nllb"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/call')"; }

case "$1" in
mfma_rate)        # profiles/r04_mfma_rate_probe.txt
  timeout 120 build/tools/mfma_rate_probe 4096 | tee $O/mfma_rate_probe.txt ;;
gemm_bench)       # profiles/r04_gemm_bench_*.txt: the big int8 GEMM forms (tuning build: CT2X_BIG_GEMM_V)
  SH="2048 14336 4096 4096 4096 4096 8192 8192 8192 2048 4096 14336 2048 6144 4096 2048 4096 4096 1024 2048 512"
  { for v in 4 1 5; do echo "== CT2X_BIG_GEMM_V=$v"; CT2X_BIG_GEMM_V=$v timeout 300 $GB $SH
      echo "== CT2X_BIG_GEMM_V=$v, swish"; CT2X_BIG_GEMM_V=$v timeout 300 $GB --act 2 2048 14336 4096; done
    echo "== edges"; timeout 300 $GB 2500 2604 640 2500 2601 384 4200 4404 384 2561 2817 256 5000 5000 512
  } 2>&1 | tee $O/gemm_bench.txt ;;
l8_prologue_ab)   # profiles/r04_l8_stream_prologue_ab.txt
  for v in 0 3 1 2 0 3; do echo "== CT2X_STREAM_PROLOGUE=$v"
    CT2X_STREAM_PROLOGUE=$v timeout 400 python tools/bench_l8.py --steps 3 --warmup 1 --no-profile 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],1),'tok/s decode_ms_per_step',round(c['decode_ms_per_step'],4),'kernels/step',c['kernels_per_decode_step'])"
  done 2>&1 | tee $O/l8_prologue_ab.txt ;;
l8_prologue_rocprof)  # profiles/r04_l8_kernel_stats_prologue_{0,3}.csv
  for v in 0 3; do
    ( cd /tmp && CT2X_STREAM_PROLOGUE=$v timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $GRAFT_REPO_ROOT/tools/bench_l8.py --steps 1 --warmup 1 --no-profile > /tmp/prof_$v.log 2>&1 )
    cp "$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)" $O/l8_kernel_stats_prologue_$v.csv
    head -10 $O/l8_kernel_stats_prologue_$v.csv | cut -c1-200
  done ;;
nllb)             # the NLLB-3.3B shape: bench line + rocprofv3 kernel stats
  timeout 900 python tools/bench_nllb.py --steps 2 --warmup 1 | tee $O/nllb.json | cut -c1-400
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_n -- python $GRAFT_REPO_ROOT/tools/bench_nllb.py --steps 1 --warmup 1 > /tmp/prof_n.log 2>&1 )
  cp "$(find /tmp/prof_n -name '*kernel_stats.csv' | head -1)" $O/nllb_kernel_stats.csv; head -12 $O/nllb_kernel_stats.csv | cut -c1-200 ;;
gemm_h_bench)     # profiles/r04_gemm_h_dma_128x128_negative.txt (CT2X_NO_GEMM_H_DMA existed in that build only)
  timeout 300 python tools/gemm_h_bench.py f16 256:2048:2048 12000:1280:1280 12000:5120:1280 1024:4096:4096 | tee $O/gemm_h_bench.txt ;;
compaction)       # tests/test_compaction_gpu.py + the reference-golden suites with rows dropped early
  timeout 900 python -m pytest tests/test_compaction_gpu.py -q -x 2>&1 | tail -5
  CT2X_COMPACT_MIN_ROWS=2 timeout 1500 python -m pytest tests/test_translator_gpu.py tests/test_generator_gpu.py tests/test_reference_kats_gpu.py tests/test_decoding_features_gpu.py tests/test_layer_variants_gpu.py tests/test_whisper_gpu.py -q -x 2>&1 | tail -4 ;;
compaction_bench) # profiles/r04_compaction_bench.txt
  { for v in 0 64 0 64; do CT2X_COMPACT_MIN_ROWS=$v timeout 600 python tools/bench_compaction.py 2>/dev/null | tail -1; done
    for v in 0 64; do BENCH_ROWS=64 CT2X_COMPACT_MIN_ROWS=$v timeout 600 python tools/bench_compaction.py 2>/dev/null | tail -1; done
  } | tee $O/compaction_bench.txt ;;
library_ab)       # profiles/r04_pinned_cache_ab.txt: build_ab/libct2_prev.so (an earlier commit's build) against this one
  { for i in 1 2 3; do
      echo "== previous library"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build_ab/libct2_prev.so timeout 300 python bench.py --steps 20 --warmup 3 $NOLEGS 2>/dev/null | line
      echo "== this build"; timeout 300 python bench.py --steps 20 --warmup 3 $NOLEGS 2>/dev/null | line
    done
    for lib in $GRAFT_REPO_ROOT/build_ab/libct2_prev.so ""; do
      echo "== beam 4, library: ${lib:-this build}"; CT2X_LIBRARY=$lib timeout 300 python bench.py --beam 4 --steps 20 --warmup 3 $NOLEGS 2>/dev/null | line
    done
  } 2>&1 | tee $O/library_ab.txt ;;
switch_ab)        # profiles/r04_fused_argmax_ab.txt, r04_session_reuse_ab.txt: gpu_r4.sh switch_ab <ENV_NAME> [bench flags]
  # e.g. switch_ab CT2X_FUSED_ARGMAX | switch_ab CT2X_SESSION_CACHE | switch_ab CT2X_GROUP_GRAPH --beam 4
  V=$2; shift 2
  { for f in 0 1 0 1; do echo "== $V=$f $*"; env $V=$f timeout 300 python bench.py --steps 20 --warmup 3 $NOLEGS "$@" 2>/dev/null | line; done
  } 2>&1 | tee $O/switch_ab_$V.txt ;;
switch_rocprof)   # per-launch times under both settings of a switch: gpu_r4.sh switch_rocprof <ENV_NAME> [bench flags]
  V=$2; shift 2
  for f in 0 1; do
    ( cd /tmp && env $V=$f timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sw$f -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 $NOLEGS "$@" > /tmp/prof_sw$f.log 2>&1 )
    cp "$(find /tmp/prof_sw$f -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_${V}_$f.csv; echo "== $V=$f"; head -10 $O/kernel_stats_${V}_$f.csv | cut -c1-160
  done ;;
beam4_rocprof)    # profiles/r04_kernel_stats_bench_tb_beam4.csv
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b4 -- python $GRAFT_REPO_ROOT/bench.py --beam 4 --steps 6 --warmup 2 $NOLEGS > /tmp/prof_b4.log 2>&1 )
  cp "$(find /tmp/prof_b4 -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_bench_tb_beam4.csv; head -20 $O/kernel_stats_bench_tb_beam4.csv | cut -c1-160 ;;
whisper)          # the Whisper-large-v3 shape
  timeout 900 python tools/bench_whisper.py --steps 2 --warmup 1 | tee $O/whisper.json | cut -c1-400 ;;
all)              # the whole -m gpu suite, smoke and the default bench line
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; tail -3 $O/pytest_all.log
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json ;;
*) echo "sections: mfma_rate gemm_bench l8_prologue_ab l8_prologue_rocprof nllb gemm_h_bench compaction compaction_bench library_ab switch_ab switch_rocprof beam4_rocprof whisper all"; exit 2 ;;
esac
