cd $GRAFT_REPO_ROOT
timeout 60 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "dequantize_gemm_output or dense_fused_vs_unfused or dense_glu" 2>&1 | tail -6
