#!/bin/bash
# Sweep of the skinny int8 GEMM configurations (NT column tiles per workgroup, waves, K-steps in
# flight per wave) at decode shapes, weights rotated through > 512 MB. Usage: skinny_sweep.sh M
M=${1:-16}
hipcc --offload-arch=gfx950 -O3 -x hip tools/gemm_bench.cpp -Iinclude -Lctranslate2_amd -lct2_mi355 -Wl,-rpath,$PWD/ctranslate2_amd -o /tmp/gb || exit 1
SHAPES="$M 6144 4096 $M 4096 4096 $M 14336 4096 $M 4096 14336 $M 128256 4096"
echo "== default dispatch"; /tmp/gb $SHAPES | grep dequant
for cfg in 1,4,4 1,4,8 1,8,4 1,8,8 1,16,4 2,4,4 2,4,8 2,8,2 2,8,4 2,8,8 2,16,2 2,16,4 4,4,2 4,4,4 4,8,2 4,8,4 4,16,2; do
  echo "== NT,WAVES,U = $cfg"; CT2X_SKINNY_CFG=$cfg /tmp/gb $SHAPES | grep dequant
done
