#!/bin/bash
# isolated warm rates of the heavy forward-form geometries through the q entry points (fp32 / q output)
for g in "8 64 256 256 128 5 1 2" "8 128 128 128 128 5 1 2" "4 256 256 256 64 3 1 1" "4 512 128 128 128 3 1 1" "4 1024 64 64 256 3 1 1" "8 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1"; do
  for q in ${2:-f32 q}; do
    echo -n "out=$q "
    python tools/conv_bench.py $g --dtype ${1:-bf16} --kinds ${3:-fwd,dgrad_t} --reps 20 --q $q | tr '\n' '|'; echo
  done
done
