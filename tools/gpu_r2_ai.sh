cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 tools/xcdp.bin > gpurun_out/r2ai_xcd_probe.txt 2>&1
tail -30 gpurun_out/r2ai_xcd_probe.txt
