cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
(for s in 4096 4224 8192 8320 16384; do timeout 120 ./build/tools/big2_loop_probe 2048 $s brief; done) > gpurun_out/r3/big2_loop_probe_strides.txt 2>&1; cat gpurun_out/r3/big2_loop_probe_strides.txt
(echo "== one wave per SIMD (gemm_s8_big2)"; ./build/tools/gemm_bench 4096 4096 4096 4096 4096 4224 8192 8192 8192 8192 8192 8320; echo "== 8-wave ping-pong (gemm_s8_big)"; CT2X_GEMM_BIG=1 ./build/tools/gemm_bench 4096 4096 4096 4096 4096 4224 8192 8192 8192 8192 8192 8320) > gpurun_out/r3/gemm_bench_strides.txt 2>&1; grep s8s8s32 gpurun_out/r3/gemm_bench_strides.txt
