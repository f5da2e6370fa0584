cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm_half_weights" > $O/m_pytest_gemm_h.log 2>&1; tail -4 $O/m_pytest_gemm_h.log
SH="256:2048:2048 256:6144:2048 256:8192:2048 256:2048:8192 12000:1280:1280 12000:5120:1280 1024:4096:4096"
{
echo "== DMA-staged form"; timeout 300 python tools/gemm_h_bench.py f16 $SH
echo "== older form (CT2X_NO_GEMM_H_DMA=1)"; CT2X_NO_GEMM_H_DMA=1 timeout 300 python tools/gemm_h_bench.py f16 $SH
echo "== DMA-staged form, bf16"; timeout 300 python tools/gemm_h_bench.py bf16 $SH
echo "== older form, bf16"; CT2X_NO_GEMM_H_DMA=1 timeout 300 python tools/gemm_h_bench.py bf16 $SH
} > $O/m_gemm_h_bench.txt 2>&1
cat $O/m_gemm_h_bench.txt
timeout 900 python tools/bench_nllb.py --steps 2 --warmup 1 > $O/m_nllb.json 2> $O/m_nllb.err; python -c "
import json; d=json.loads(open('$O/m_nllb.json').read().strip().splitlines()[-1]); print('NLLB', d['value'], d['config']['decode_ms_per_step'], d['config']['kernels_per_decode_step'])"
