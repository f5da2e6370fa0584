cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "norm_gemm" 2>&1 | tail -2
for s in 1 2 4 8; do
  echo "== eager streams=$s"; timeout 300 python bench.py --no-cpu-baseline --no-l8 --no-profile --streams $s 2>/dev/null | cut -c1-190
done
for s in 2 4 8; do
  echo "== graphs streams=$s"; CT2X_STREAM_GRAPHS=1 timeout 300 python bench.py --no-cpu-baseline --no-l8 --no-profile --streams $s 2>&1 | tail -2 | cut -c1-300
done
