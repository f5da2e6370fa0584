cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 900 python -m pytest tests/test_compaction_gpu.py -q -x > $O/q_pytest_compaction.log 2>&1; tail -25 $O/q_pytest_compaction.log
CT2X_COMPACT_MIN_ROWS=2 timeout 1500 python -m pytest tests/test_translator_gpu.py tests/test_generator_gpu.py tests/test_reference_kats_gpu.py tests/test_decoding_features_gpu.py tests/test_layer_variants_gpu.py tests/test_whisper_gpu.py -q -x > $O/q_pytest_goldens_forced_compaction.log 2>&1; tail -6 $O/q_pytest_goldens_forced_compaction.log
