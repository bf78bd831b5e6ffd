#!/bin/bash
# heavy geometries of the joint step (profiles/r01_kernel_table.txt) through the bf16 matrix-core kernels, next to fp32
for g in "8 64 256 256 128 5 1 2" "8 128 128 128 128 5 1 2" "4 256 256 256 64 3 1 1" "4 512 128 128 128 3 1 1" \
         "4 1024 64 64 256 3 1 1" "4 1024 32 32 512 3 1 1" "4 64 128 128 256 3 1 1" "8 64 256 256 128 3 2 1" \
         "8 128 128 128 256 3 2 1" "8 128 64 64 128 5 1 2"; do
  python tools/conv_bench.py $g --dtype ${1:-bf16} --kinds fwd,dgrad_t,wgrad --reps 10
  if [ -n "$2" ]; then python tools/conv_bench.py $g --kinds fwd,dgrad_t,wgrad --reps 10; fi
done
