cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r2zzz_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r2zzz_pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-profile > gpurun_out/r2zzz_bench.json 2> gpurun_out/r2zzz_bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r2zzz_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('kernels_per_decode_step')); print(json.dumps(d['config'].get('l8'))[:420])"
