cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -x hip tools/norm_probe.cpp -Ictranslate2_amd/csrc -Iinclude -o /tmp/np 2>/dev/null && /tmp/np | tee gpurun_out/r2m_norm_probe.txt
for i in 1 2; do
for f in 1 0; do
  echo "== NORM_FUSION=$f"; CT2X_NORM_FUSION=$f timeout 300 python bench.py --no-cpu-baseline --no-l8 --no-profile 2>/dev/null | cut -c1-190
done
done
