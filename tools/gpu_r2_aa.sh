cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_translator_gpu.py tests/test_generator_gpu.py tests/test_reference_kats_gpu.py tests/test_tensor_parallel_gpu.py -q -x > gpurun_out/r2aa_parity.log 2>&1; tail -3 gpurun_out/r2aa_parity.log
for i in 1 2; do
for a in 1 0; do echo "== ANCESTRY=$a"; CT2X_BEAM_ANCESTRY=$a python bench.py --beam 4 --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | cut -c1-190; done
done
