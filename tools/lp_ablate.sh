#!/bin/bash
# what bounds the low-precision forward-form kernel: q operands, fp32 / q output, and the GHM_ABLATE bits
# (1 no patch loads, 2 no MFMAs, 4 no stores, 8 no weight loads); results of ablated runs are wrong by construction
for g in "8 64 256 256 128 5 1 2" "4 256 256 256 64 3 1 1" "4 512 128 128 128 3 1 1" "8 64 256 256 128 3 2 1"; do
  for q in f32 q; do
    for ab in 0 1 8 9 2 4 6; do
      echo -n "out=$q ablate=$ab  "
      GHM_ABLATE=$ab python tools/conv_bench.py $g --dtype ${1:-bf16} --kinds fwd --reps 20 --q $q | tail -1
    done
  done
done
