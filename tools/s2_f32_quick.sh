#!/bin/bash
# isolated warm rates of the fp32 3x3 stride-2 family (conv_patch<3,..,2,2>, dgrad_s2_patch_kernel, wgrad_patch<3,2,..>) on the
# U-Net encoder / PatchGAN geometries; dgrad per tile shape (GHM_DGRAD_S2_TILE: 0 = 128 ch x 2 rows, 1 = 64 x 4, 2 = 64 x 2)
for g in "8 64 256 256 128 3 2 1" "4 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1" "4 128 128 128 256 3 2 1" "8 256 64 64 512 3 2 1" "4 256 64 64 512 3 2 1"; do
  echo "== $g"
  python tools/conv_bench.py $g --kinds fwd,wgrad --reps 50 | tr '\n' '|'; echo
  for t in "" 0 1 2; do
    echo -n "dgrad tile=${t:-plan} "
    env ${t:+GHM_DGRAD_S2_TILE=$t} python tools/conv_bench.py $g --kinds dgrad_t --reps 50 | tr '\n' '|'; echo
  done
done
