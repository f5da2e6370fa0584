cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 900 python -m pytest tests/test_decoding_features_gpu.py -q -s > gpurun_out/r3/pytest_features.log 2>&1; tail -25 gpurun_out/r3/pytest_features.log
