#!/bin/bash
# isolated warm rates of the 3x3 stride-2 data gradient (lp_dgrad_s2_kernel) on the U-Net encoder / PatchGAN geometries,
# per tile shape (GHM_LP_DGRAD_S2_TILE: 0 = 128 ch x 2 rows, 1 = 64 x 4, 2 = 64 x 2; unset = the plan's choice)
for g in "8 64 256 256 128 3 2 1" "4 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1" "8 256 64 64 512 3 2 1" "4 512 32 32 512 3 2 1"; do
  for t in "" 0 1 2; do
    echo -n "tile=${t:-plan} "
    env ${t:+GHM_LP_DGRAD_S2_TILE=$t} python tools/conv_bench.py $g --dtype ${1:-bf16} --kinds dgrad_t --reps 50 --q ${2:-both} | tr '\n' '|'; echo
  done
done
