#!/bin/bash
# small-map layers of the step: the conv_small.hip path against what served them before (GHM_NO_SM), isolated and warm
#   gpurun -- 'bash tools/sm_sweep.sh'
for g in "4 512 32 32 512 3 2 1" "4 512 16 16 512 3 2 1" "4 512 8 8 512 3 2 1" "4 512 4 4 512 3 2 1" "4 512 2 2 512 2 1 0" \
         "4 1024 4 4 512 3 1 1" "4 1024 8 8 512 3 1 1" "4 1024 16 16 512 3 1 1" "4 512 4 4 256 5 1 2" "4 256 4 4 1024 3 1 1" \
         "4 256 8 8 512 3 1 1" "4 128 16 16 512 3 1 1" "8 256 16 16 256 5 1 2" "8 256 8 8 256 5 1 2"; do
  for sw in "" "GHM_NO_SM=1"; do
    echo "== $g ${sw:-sm}"
    env $sw python tools/conv_bench.py $g --dtype bf16 --q both --kinds fwd,dgrad_t --reps 100 2>&1 | grep -v "^$" | head -4
  done
done
