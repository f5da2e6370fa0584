cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/ct2data/models/v2
ln -sfn $GRAFT_REPO_ROOT/tests/golden/models/aren-transliteration /tmp/ct2data/models/v2/aren-transliteration
ln -sfn $GRAFT_REPO_ROOT/tests/golden/models/aren-transliteration-i8 /tmp/ct2data/models/v2/aren-transliteration-i8
export CT2_USE_MKL=1
timeout 900 oracle/_ref/ct2_reftests /tmp/ct2data --gtest_filter='CUDA/*-*float16*:*bfloat16*' > gpurun_out/r2ad_reftests_cuda.log 2>&1; echo "rc=$?"; tail -45 gpurun_out/r2ad_reftests_cuda.log
