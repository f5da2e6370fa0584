cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_generator_gpu.py -q -s -k "teacher_forced or generate_matches or ragged_prompts_match" > gpurun_out/r2v_gen.log 2>&1; grep -n "int8:\|identical\|passed\|failed\|FAILED" gpurun_out/r2v_gen.log | head -40
