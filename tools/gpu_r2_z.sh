cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_dbg
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -x hip tools/kernel_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/kb 2>/dev/null
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_dbg -- /tmp/kb 32 > /tmp/o.txt 2> /tmp/e.txt); tail -5 /tmp/e.txt
find gpurun_out/pmc_dbg -type f | head; f=$(find gpurun_out/pmc_dbg -name "*counter*" | head -1); head -3 "$f"; wc -l "$f"
find gpurun_out/pmc_dbg -name "*.csv" -size +1M -delete; find gpurun_out/pmc_dbg -name "*.db" -delete
