#!/bin/bash
# Round 6: the GPU measurements of the round as named sections (one `gpurun` call each):
#   gpurun --timeout 900 -- 'bash tools/gpu_r6.sh <section>'
# Output goes to gpurun_out/r6/; what is kept is copied to profiles/ (profiles/README.md).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
O=gpurun_out/r6
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
export TMPDIR=/tmp
GB=build/tools/gemm_bench
GBT=build/tools/gemm_bench_tuning
NOLEGS="--no-cpu-baseline --no-profile --no-l8 --no-whisper --no-nllb"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/call', 'kernels/step', d['config'].get('kernels_per_decode_step'))"; }
L8SH="2048 14336 4096"

case "$1" in
pmc_big3)         # profiles/r06_pmc_gemm_big3.json: which unit the MFMA-bound GEMM keeps busy
  timeout 900 python tools/pmc_gemm_big3.py $O/pmc_big3 > $O/pmc_gemm_big3.json 2> $O/pmc_big3.err
  head -c 6000 $O/pmc_gemm_big3.json; tail -3 $O/pmc_big3.err
  find $O/pmc_big3 -name "*.csv" -size +1M -delete; find $O/pmc_big3 -name "*.db" -delete
  { echo "== shipped library"; timeout 300 $GB --act 2 $L8SH; timeout 300 $GB $L8SH 4096 4096 4096 8192 8192 8192; } 2>&1 | tee $O/gemm_bench_base.txt ;;
gemm_bench)       # the big int8 GEMM forms, tuning build; extra args = env assignments to A/B
  shift
  { echo "== default"; timeout 300 $GBT --act 2 $L8SH; timeout 300 $GBT $L8SH 4096 4096 4096 8192 8192 8192 2048 6144 4096 2048 4096 14336 1024 2048 512
    for e in "$@"; do echo "== $e"; env $e timeout 300 $GBT --act 2 $L8SH; env $e timeout 300 $GBT $L8SH 4096 4096 4096 8192 8192 8192; done
  } 2>&1 | tee $O/gemm_bench.txt ;;
gemm_ab)          # head (round 5) library vs this tree, then the tile shapes (tuning build)
  SH="$L8SH 4096 4096 4096 8192 8192 8192 2048 6144 4096 2048 4096 4096 2048 4096 14336 1024 2048 512"
  { echo "== head (round-5 kernel), swish"; timeout 300 build/tools/gemm_bench_head --act 2 $L8SH 2048 6144 4096
    echo "== this tree, swish"; timeout 300 $GB --act 2 $L8SH 2048 6144 4096
    echo "== head, relu"; timeout 300 build/tools/gemm_bench_head $SH
    echo "== this tree, relu"; timeout 300 $GB $SH
    for sh in 0 1 2; do echo "== tuning CT2X_BIG3_SHAPE=$sh swish"; CT2X_BIG3_SHAPE=$sh timeout 300 $GBT --act 2 $L8SH 4096 4096 4096; done
    for sh in 0 1 2; do echo "== tuning CT2X_BIG3_SHAPE=$sh CT2X_BIG_GEMM_V=3 (int32-out on this kernel too)"; CT2X_BIG_GEMM_V=3 CT2X_BIG3_SHAPE=$sh timeout 300 $GBT $L8SH 4096 4096 4096; done
    echo "== tuning K loops only (CT2X_BIG3_VAR=2) shapes 0 1"; for sh in 0 1; do CT2X_BIG_GEMM_V=3 CT2X_BIG3_VAR=2 CT2X_BIG3_SHAPE=$sh timeout 300 $GBT $L8SH; done
    echo "== edges, this tree"; timeout 300 $GB 2500 2604 640 2500 2601 384 4200 4404 384 2561 2817 256 5000 5000 512
  } 2>&1 | tee $O/gemm_ab.txt ;;
gemm_ab2)         # the round's first kernel (build/libct2_mi355_r6a.so: epilogue + tile widths) vs this tree (deep ring)
  SH="$L8SH 4096 4096 4096 8192 8192 8192 2048 6144 4096 2048 4096 4096 2048 4096 14336 1024 2048 512"
  { echo "== r6a, swish"; timeout 300 build/tools/gemm_bench_r6a --act 2 $L8SH 2048 6144 4096 4096 4096 4096
    echo "== this tree, swish"; timeout 300 $GB --act 2 $L8SH 2048 6144 4096 4096 4096 4096
    echo "== r6a, relu"; timeout 300 build/tools/gemm_bench_r6a $SH
    echo "== this tree, relu"; timeout 300 $GB $SH
    for sh in 0 1; do echo "== tuning CT2X_BIG3_SHAPE=$sh CT2X_BIG_GEMM_V=3 (int32-out on this kernel too), then K loops only"; CT2X_BIG_GEMM_V=3 CT2X_BIG3_SHAPE=$sh timeout 300 $GBT $L8SH 4096 4096 4096 8192 8192 8192
      CT2X_BIG_GEMM_V=3 CT2X_BIG3_VAR=2 CT2X_BIG3_SHAPE=$sh timeout 300 $GBT $L8SH 4096 4096 4096; done
    echo "== edges, this tree"; timeout 300 $GB 2500 2604 640 2500 2601 384 4200 4404 384 2561 2817 256 5000 5000 512
  } 2>&1 | tee $O/gemm_ab2.txt ;;
gemm_ab3)         # r6a (epilogue + tile widths) vs this tree (+ the all-interior build of the ReLU / plain forms)
  SH="$L8SH 4096 4096 4096 8192 8192 8192 2048 6144 4096 2048 4096 4096 2048 4096 14336 1024 2048 512"
  { echo "== r6a, relu"; timeout 300 build/tools/gemm_bench_r6a $SH
    echo "== this tree, relu"; timeout 300 $GB $SH
    echo "== this tree, swish"; timeout 300 $GB --act 2 $L8SH 2048 6144 4096 4096 4096 4096
    echo "== this tree, no activation"; timeout 300 $GB --act -1 $L8SH 4096 4096 4096
  } 2>&1 | tee $O/gemm_ab3.txt ;;
gemm_nt)          # the result leaving with the non-temporal policy (build/libct2_mi355_nt.so) vs the default, alternating
  { for i in 1 2; do echo "== default"; timeout 300 $GB --act 2 $L8SH; timeout 300 $GB $L8SH 4096 4096 4096 2048 6144 4096
      echo "== nt stores"; timeout 300 build/tools/gemm_bench_nt --act 2 $L8SH; timeout 300 build/tools/gemm_bench_nt $L8SH 4096 4096 4096 2048 6144 4096; done
  } 2>&1 | tee $O/gemm_nt.txt ;;
beam_fused)       # beam search: LogSoftMax + per-row candidates, merge + select + advance (CT2X_BEAM_FUSED_TOPK)
  timeout 1200 python -m pytest tests/test_translator_gpu.py tests/test_reference_kats_gpu.py tests/test_decoding_features_gpu.py -x -q -k "beam or Beam or prefix or alternatives or bias" 2>&1 | tail -5 | tee $O/beam_fused_tests.txt
  { for f in 0 1 0 1; do echo "== CT2X_BEAM_FUSED_TOPK=$f --beam 4"; CT2X_BEAM_FUSED_TOPK=$f timeout 300 python bench.py --beam 4 --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
  } 2>&1 | tee $O/beam_fused_ab.txt ;;
bench_full)       # the default bench line, as the driver runs it
  timeout 1400 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 400 $O/bench_full.err
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r6/bench_full.json"))
c = d["config"]
print("value", round(d["value"], 1), "ms_per_step", round(d["ms_per_step"], 3), "kernels/step", c.get("kernels_per_decode_step"))
print("beam4", c.get("beam4"))
print("prefill", {k: (round(v.get("avg_us", 0), 1), round(v.get("roofline", {}).get("frac", 0), 4)) for k, v in c.get("prefill_gemm", {}).items() if isinstance(v, dict)})
print("ragged", json.dumps(c.get("ragged_sources"))[:1200])
for leg in ("l8", "nllb", "whisper", "tp_local", "tp_shm", "no_session_reuse"):
    v = c.get(leg)
    print(leg, (json.dumps(v)[:300] if v is not None else None))
print("roofline", json.dumps(d.get("roofline"))[:600])
print("cpu", d.get("cpu_baseline"))
print("tokens_equal_unfused", c.get("tokens_equal_unfused"))
PY
  ;;
cluster_stamps)   # phase timeline of workgroup 0 of the decoder-stack launch (last step), incl. the attention core's phases
  CT2X_CLUSTER_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 1 $NOLEGS --no-token-check 2>&1 | grep -A8 "decoder_stack timeline" | tail -9 | tee $O/cluster_timeline.txt
  { for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 3 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done; } | tee $O/cluster_speed.txt ;;
cluster_check)    # the decoder-stack launch's own tests, then the timeline and the speed
  timeout 900 python -m pytest tests/test_decoder_cluster_gpu.py tests/test_bench_path_gpu.py -x -q 2>&1 | tail -4 | tee $O/cluster_tests.txt
  CT2X_CLUSTER_STAMPS=1 timeout 300 python bench.py --steps 2 --warmup 1 $NOLEGS --no-token-check 2>&1 | grep -A8 "decoder_stack timeline" | tail -9 | tee $O/cluster_timeline.txt
  { for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done; } | tee $O/cluster_speed.txt ;;
trace_gaps)       # profiles/r06_tb_kernel_trace_gaps.txt + r06_kernel_stats_bench_tb_b32.csv: rocprofv3 kernel trace of the headline leg
  cd /tmp; rm -rf /tmp/r6trace; ( cd $GRAFT_REPO_ROOT; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r6trace -- python bench.py --steps 12 --warmup 3 $NOLEGS --no-token-check > $O/trace_bench.log 2>&1 )
  cd $GRAFT_REPO_ROOT
  f=$(find /tmp/r6trace -name "*kernel_trace.csv" | head -1); python tools/trace_gaps.py "$f" 2>&1 | head -40 | tee $O/tb_kernel_trace_gaps.txt
  g=$(find /tmp/r6trace -name "*kernel_stats.csv" | head -1); cp "$g" $O/kernel_stats_bench_tb_b32.csv; head -8 "$g" | cut -c1-200
  tail -1 $O/trace_bench.log | cut -c1-300 ;;
rows256)          # the 129 .. 256-row 16-bit GEMM: its tests, then the NLLB leg with CT2X_H_ROWS256=0 / 1 alternating
  timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm_half" 2>&1 | tail -3 | tee $O/rows256_tests.txt
  timeout 900 python -m pytest tests/test_translator_gpu.py -x -q -k "nllb or half" 2>&1 | tail -3 | tee -a $O/rows256_tests.txt
  { for f in 0 1 0 1; do echo "== CT2X_H_ROWS256=$f"; CT2X_H_ROWS256=$f timeout 600 python tools/bench_nllb.py --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', d['config'].get('decode_ms_per_step'), 'ms/step', d['config'].get('kernels_per_decode_step'))"; done; } 2>&1 | tee $O/rows256_ab.txt ;;
rows256_kernels)  # per-kernel times of the 256-row 16-bit GEMM forms (rocprofv3 --stats), NLLB decode shapes
  SH="256:8192:2048 256:2048:8192 256:2048:2048 256:6144:2048"
  for f in 0 1; do
    rm -rf /tmp/r256p$f; ( cd /tmp; CT2X_H_ROWS256=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r256p$f -- python $GRAFT_REPO_ROOT/tools/gemm_h_bench.py ${KIND:-f16} $SH > $GRAFT_REPO_ROOT/$O/rows256_bench_$f.txt 2>&1 )
    echo "== CT2X_H_ROWS256=$f"; grep -v "^W20\|rocprof" $O/rows256_bench_$f.txt | tail -6
    g=$(find /tmp/r256p$f -name "*kernel_stats.csv" | head -1); cp "$g" $O/rows256_kernel_stats_$f.csv; head -8 "$g" | cut -c1-60,150-260
  done ;;
leader)           # the streaming Dense's leader prologues: op tests, Llama-3 geometry goldens, then the L8 leg A/B
  timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "streaming or stream" 2>&1 | tail -4 | tee $O/leader_tests.txt
  timeout 900 python -m pytest tests/test_generator_gpu.py tests/test_bench_path_gpu.py -x -q -k "llama3 or Llama" 2>&1 | tail -4 | tee -a $O/leader_tests.txt
  for v in ${MASKS:-0 12 4 8 0 12}; do echo "== CT2X_STREAM_PROLOGUE=$v"
    CT2X_STREAM_PROLOGUE=$v timeout 400 python tools/bench_l8.py --steps 3 --warmup 1 --no-profile 2>&1 | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],1),'tok/s decode_ms_per_step',round(c['decode_ms_per_step'],4),'kernels/step',c['kernels_per_decode_step'])"
  done 2>&1 | tee $O/l8_leader_ab.txt ;;
leader_rocprof)   # per-kernel times of the L8 leg with and without the leader forms
  for v in 0 12; do
    ( cd /tmp && CT2X_STREAM_PROLOGUE=$v timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $GRAFT_REPO_ROOT/tools/bench_l8.py --steps 1 --warmup 1 --no-profile > /tmp/prof_$v.log 2>&1 )
    cp "$(find /tmp/prof_$v -name '*kernel_stats.csv' | head -1)" $O/l8_kernel_stats_leader_$v.csv
    python - $O/l8_kernel_stats_leader_$v.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r['Name'][:110].ljust(110), r['Calls'], round(float(r['AverageNs']) / 1e3, 2), r['Percentage'])
PY
  done ;;
chain)            # the ordered chains (norm statistics, exp sums) as two batches of reads swapping roles
  timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_primitives_gpu.py -x -q -k "norm or softmax or Norm or SoftMax" 2>&1 | tail -3 | tee $O/chain_tests.txt
  timeout 1200 python -m pytest tests/test_translator_gpu.py tests/test_reference_kats_gpu.py tests/test_generator_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/chain_tests.txt
  { for i in 1 2; do echo "== --beam 4"; timeout 300 python bench.py --beam 4 --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
    echo "== greedy"; timeout 300 python bench.py --steps 10 --warmup 2 $NOLEGS 2>&1 | tail -1 | line
  } 2>&1 | tee $O/chain_bench.txt ;;
leg_stats)        # rocprofv3 kernel stats of the legs: profiles/r06_final_kernel_stats_bench_{l8,nllb,whisper,tb_beam4}.csv
  show() { python - "$1" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:105].ljust(105), r['Calls'].rjust(6), str(round(float(r['AverageNs']) / 1e3, 2)).rjust(8), r['Percentage'])
PY
  }
  for leg in ${LEGS:-nllb whisper beam4 l8}; do
    case $leg in
      nllb)    cmd="tools/bench_nllb.py --steps 1 --warmup 1" ;;
      whisper) cmd="tools/bench_whisper.py --steps 1 --warmup 1" ;;
      l8)      cmd="tools/bench_l8.py --steps 1 --warmup 1 --no-profile" ;;
      beam4)   cmd="bench.py --beam 4 --steps 5 --warmup 2 $NOLEGS --no-token-check" ;;
    esac
    rm -rf /tmp/prof_$leg
    ( cd /tmp && timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$leg -- python $GRAFT_REPO_ROOT/$cmd > /tmp/prof_$leg.log 2>&1 )
    f=$(find /tmp/prof_$leg -name '*kernel_stats.csv' | head -1)
    echo "== $leg"; tail -1 /tmp/prof_$leg.log | cut -c1-200
    [ -n "$f" ] && cp "$f" $O/final_kernel_stats_bench_$leg.csv && show $O/final_kernel_stats_bench_$leg.csv
  done ;;
awq_bench)        # decode-row AWQ launches, both 16-bit arms
  { timeout 300 python tools/awq_bench.py 16; timeout 300 python tools/awq_bench.py 16 bf16; } 2>&1 | grep "^awq" | tee $O/awq_bench.txt ;;
rows128)          # the 128-row form of the 129 .. 256-row 16-bit GEMM: tests (both forms), launches, NLLB leg
  timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm_half" 2>&1 | tail -2 | tee $O/rows128_tests.txt
  CT2X_H_ROWS256=1 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm_half" 2>&1 | tail -2 | tee -a $O/rows128_tests.txt
  timeout 900 python -m pytest tests/test_translator_gpu.py -x -q -k "nllb or half" 2>&1 | tail -2 | tee -a $O/rows128_tests.txt
  SH="256:2048:2048 256:6144:2048 256:8192:2048 256:2048:8192"
  { for cfg in "CT2X_H_ROWS256=1" "CT2X_H_ROWS256=2" "CT2X_H_ROWS256=2 CT2X_H_ROWS_TARGET=512" "CT2X_H_ROWS256=2 CT2X_H_ROWS_TARGET=128"; do
      echo "== $cfg"; env $cfg timeout 300 python tools/gemm_h_bench.py ${KIND:-f16} $SH 2>&1 | grep "per launch"; done
    echo "== bf16, CT2X_H_ROWS256=1 then 2"; for m in 1 2; do CT2X_H_ROWS256=$m timeout 300 python tools/gemm_h_bench.py bf16 $SH 2>&1 | grep "per launch"; done
  } 2>&1 | tee $O/rows128_bench.txt
  { for cfg in "CT2X_H_ROWS256=1" "CT2X_H_ROWS256=2" "CT2X_H_ROWS256=2 CT2X_H_ROWS_TARGET=512" "CT2X_H_ROWS256=1" "CT2X_H_ROWS256=2"; do echo "== $cfg"; env $cfg timeout 600 python tools/bench_nllb.py --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', d['config'].get('decode_ms_per_step'), 'ms/step', d['config'].get('kernels_per_decode_step'))"; done; } 2>&1 | tee $O/rows128_ab.txt ;;
beam_fast)        # 16-bit beam search: candidates instead of rows (CT2X_BEAM_FUSED_FAST): A/B test, NLLB + Whisper legs
  timeout 1200 python -m pytest tests/test_beam_fused_fast_gpu.py -x -q 2>&1 | tail -3 | tee $O/beam_fast_tests.txt
  timeout 1200 python -m pytest tests/test_translator_gpu.py tests/test_whisper_gpu.py -x -q -k "half or nllb or float16 or bfloat16 or beam" 2>&1 | tail -3 | tee -a $O/beam_fast_tests.txt
  { for f in 0 1 0 1; do echo "== nllb CT2X_BEAM_FUSED_FAST=$f"; CT2X_BEAM_FUSED_FAST=$f timeout 600 python tools/bench_nllb.py --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', d['config'].get('decode_ms_per_step'), 'ms/step', d['config'].get('kernels_per_decode_step'))"; done
    for f in 0 1; do echo "== whisper CT2X_BEAM_FUSED_FAST=$f"; CT2X_BEAM_FUSED_FAST=$f timeout 600 python tools/bench_whisper.py --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-260; done
  } 2>&1 | tee $O/beam_fast_ab.txt ;;
rows128_big)      # more than 256 rows on the 128-row form (CT2X_H_ROWS128_BIG): tests, launches, Whisper leg
  timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm_half" 2>&1 | tail -2 | tee $O/rows128_big_tests.txt
  timeout 900 python -m pytest tests/test_whisper_gpu.py tests/test_translator_gpu.py -x -q -k "half or float16 or bfloat16 or nllb or whisper" 2>&1 | tail -2 | tee -a $O/rows128_big_tests.txt
  SH="12000:1280:1280 12000:3840:1280 12000:5120:1280 12000:1280:5120 2048:8192:2048 2048:2048:8192"
  { for f in 0 1; do for kind in bf16 f16; do echo "== CT2X_H_ROWS128_BIG=$f $kind"; CT2X_H_ROWS128_BIG=$f timeout 300 python tools/gemm_h_bench.py $kind $SH 2>&1 | grep "per launch"; done; done
  } 2>&1 | tee $O/rows128_big_bench.txt
  { for f in 0 1 0 1; do echo "== whisper CT2X_H_ROWS128_BIG=$f"; CT2X_H_ROWS128_BIG=$f timeout 600 python tools/bench_whisper.py --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],1), 'tok/s', {k: (round(v,2) if isinstance(v,float) else v) for k,v in c.items() if 'ms' in k})"; done
    for f in 0 1; do echo "== nllb CT2X_H_ROWS128_BIG=$f"; CT2X_H_ROWS128_BIG=$f timeout 600 python tools/bench_nllb.py --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', d['config'].get('decode_ms_per_step'), 'ms/step', d['ms_per_step'])"; done
  } 2>&1 | tee $O/rows128_big_ab.txt ;;
pmc)              # profiles/r06_pmc_traffic.json: memory-side bytes per launch of the headline leg's kernels
  timeout 600 python tools/pmc_bench.py $O/pmc > $O/pmc_traffic.json; head -30 $O/pmc_traffic.json; find $O/pmc -name "*.csv" -size +1M -delete; find $O/pmc -name "*.db" -delete ;;
final)            # the round's closing evidence in one call: bench line, kernel trace of the headline leg, leg stats, PMC traffic
  bash tools/gpu_r6.sh bench_full
  bash tools/gpu_r6.sh trace_gaps
  LEGS="l8 nllb whisper beam4" bash tools/gpu_r6.sh leg_stats > $O/leg_stats.txt 2>&1; tail -70 $O/leg_stats.txt | cut -c1-150
  bash tools/gpu_r6.sh pmc ;;
beam_proj_sweep)  # beam 4: the 128-row vocabulary projection's row tiles per workgroup (tuning build, CT2X_SKINNY_MT)
  { for mt in 1 2 4 8 1 2 4 8; do echo "== CT2X_SKINNY_MT=$mt --beam 4"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/ctranslate2_amd/libct2_mi355_tuning.so CT2X_SKINNY_MT=$mt timeout 300 python bench.py --beam 4 --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
  } 2>&1 | tee $O/beam_proj_sweep.txt ;;
beam_tiles)       # beam search's decoder-stack launch: four K / V tiles, one attention round (build/libct2_mi355_new*.so) vs two tiles, two rounds (_old.so)
  timeout 900 python -m pytest tests/test_decoder_cluster_gpu.py tests/test_bench_path_gpu.py -x -q 2>&1 | tail -4 | tee $O/beam_tiles_tests.txt
  { for v in old new new2 old new new2; do echo "== $v --beam 4"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so timeout 300 python bench.py --beam 4 --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
    for v in old new new2; do echo "== $v --beam 2"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so timeout 300 python bench.py --beam 2 --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
    for v in old new new2; do echo "== $v --batch 128 (greedy, several rows per cluster)"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so timeout 300 python bench.py --batch 128 --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
    for v in old new new2; do echo "== $v --batch 64 (greedy, two rows per cluster)"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so timeout 300 python bench.py --batch 64 --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
  } 2>&1 | tee $O/beam_tiles_ab.txt
  for v in old new new2; do echo "== timeline $v"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so CT2X_CLUSTER_STAMPS=1 timeout 300 python bench.py --beam 4 --steps 2 --warmup 1 $NOLEGS --no-token-check 2>&1 | grep -A8 "decoder_stack timeline" | tail -9 | head -8 | cut -c1-110; done | tee $O/beam_tiles_timeline.txt ;;
proj128)          # the 128-row vocabulary projection of beam 4 ([128, 32000] x 512): skinny row tiles / the MFMA tile kernels (tuning build)
  { for e in "X=1" "CT2X_SKINNY_MT=2" "CT2X_SKINNY_MT=4" "CT2X_SKINNY_MT=8" "CT2X_NO_SKINNY_GEMM=1" "CT2X_NO_SKINNY_GEMM=1 CT2X_NO_BIG_GEMM=1"; do echo "== $e"; env $e timeout 300 $GBT --act -1 128 32000 512 64 32000 512 32 32000 512 2>&1 | tail -8; done
  } 2>&1 | tee $O/proj128.txt ;;
padder)           # Padder: its test, then the ragged-source leg with CT2X_PADDER=0 / 1 (batches of 32 and 256 sentences)
  timeout 900 python -m pytest tests/test_padder_gpu.py -x -q 2>&1 | tail -5 | tee $O/padder_tests.txt
  { for f in 0 1 0 1; do echo "== CT2X_PADDER=$f"; CT2X_PADDER=$f timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-l8 --no-whisper --no-nllb --no-token-check 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['config'].get('ragged_sources', {})
for k in ('batch32', 'batch256'):
    e = r.get(k, {})
    print(k, 'ragged', round(e['ragged']['ms_per_call'], 3), 'ms/call, encode', round(e['ragged']['encode_ms'], 3), 'ms | full length', round(e['full_length']['ms_per_call'], 3), 'encode', round(e['full_length']['encode_ms'], 3), '| padded share', round(e['padded_row_share'], 3))
"; done; } 2>&1 | tee $O/padder_ab.txt ;;
act_once)         # the tiled GEMMs' epilogues with the activation as a constant (build/libct2_mi355_new.so) vs decided per element (_old.so)
  timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -2 | tee $O/act_once_tests.txt
  SH="12000:1280:1280 12000:3840:1280 12000:5120:1280 12000:1280:5120 2048:8192:2048"
  { for v in old new old new; do echo "== $v bf16"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so timeout 300 python tools/gemm_h_bench.py bf16 $SH 2>&1 | grep "per launch"; done
    for v in old new old new; do echo "== whisper $v"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so timeout 600 python tools/bench_whisper.py --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['value'],1), 'tok/s', {k: c[k] for k in c if 'ms' in k})"; done
    for v in old new old new; do echo "== headline $v"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so timeout 300 python bench.py --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'encode', d['config'].get('encode_ms_last_call'))"; done
  } 2>&1 | tee $O/act_once_ab.txt ;;
gemv_chains)      # the cluster launches' GEMVs on two integer chains per thread (build/libct2_mi355_new.so) vs one (_old.so)
  timeout 900 python -m pytest tests/test_decoder_cluster_gpu.py tests/test_bench_path_gpu.py -x -q 2>&1 | tail -2 | tee $O/gemv_chains_tests.txt
  { for v in old new old new old new; do echo "== $v headline"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so timeout 300 python bench.py --steps 20 --warmup 3 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
    for v in old new old new; do echo "== $v --beam 4"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so timeout 300 python bench.py --beam 4 --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
    for v in old new; do echo "== $v --batch 128"; CT2X_LIBRARY=$GRAFT_REPO_ROOT/build/libct2_mi355_$v.so timeout 300 python bench.py --batch 128 --steps 10 --warmup 2 $NOLEGS --no-token-check 2>&1 | tail -1 | line; done
  } 2>&1 | tee $O/gemv_chains_ab.txt ;;
suite_rest)       # the -m gpu suite from a given test file on (after a fix)
  shift; timeout 2400 python -m pytest "$@" -m gpu -x -q > $O/pytest_rest.log 2>&1; tail -5 $O/pytest_rest.log ;;
tests)            # a list of test files / pytest args
  shift; timeout 1500 python -m pytest "$@" -x -q 2>&1 | tail -15 | tee $O/tests.txt ;;
suite)            # the whole -m gpu suite only
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; tail -5 $O/pytest_all.log ;;
*) echo "unknown section $1"; exit 2 ;;
esac
