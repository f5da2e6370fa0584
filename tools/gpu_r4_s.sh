cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
O=gpurun_out/r4
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/ctranslate2_amd:$LD_LIBRARY_PATH
timeout 1500 python -m pytest tests -m gpu -x -q > $O/s_pytest_all.log 2>&1; tail -5 $O/s_pytest_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $O/s_bench.json 2> $O/s_bench.err; cut -c1-400 $O/s_bench.json
