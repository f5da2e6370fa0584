cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(echo "== L8 shape rows=16 heads=32/8 dh=128 t_max=256 len=192 (per-q-head split mode)"; python tools/attn_trace.py 16 32 8 128 256 192;
 echo "== len=255"; python tools/attn_trace.py 16 32 8 128 256 255;
 echo "== CT2X_ATTN_SHARE_KV=1 len=192"; CT2X_ATTN_SHARE_KV=1 python tools/attn_trace.py 16 32 8 128 256 192) > gpurun_out/r2as_attn_trace.txt 2>&1
cat gpurun_out/r2as_attn_trace.txt
