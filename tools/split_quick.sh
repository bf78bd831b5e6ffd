#!/bin/bash
# fp32 by operand splitting (conv_split.hip) beside the fp32 MFMA kernels: isolated warm rates on the step's big layers
for g in "8 128 128 128 128 5 1 2" "8 64 256 256 64 5 1 2" "4 256 128 128 128 3 1 1" "4 128 256 256 64 3 1 1" "8 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1"; do
  echo "== $g"
  python tools/conv_bench.py $g --kinds fwd,dgrad_t,wgrad --reps 30 | tr '\n' '|'; echo
  python tools/conv_bench.py $g --kinds fwd,dgrad_t,pack_x --reps 30 --dtype split | tr '\n' '|'; echo
  python tools/conv_bench.py $g --kinds fwd,dgrad_t,wgrad --reps 30 --dtype split --q q | tr '\n' '|'; echo
  python tools/conv_bench.py $g --kinds fwd,dgrad_t,wgrad --reps 30 --dtype split2 --q q | tr '\n' '|'; echo      # two pieces, three products (bf16x2)
done
