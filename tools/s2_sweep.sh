# tuning sweep: split-K policy of the low-precision forward / stride-1 data-gradient kernel
G=("4 1024 64 64 256 3 1 1" "4 1024 32 32 512 3 1 1" "4 512 128 128 128 3 1 1" "4 256 64 64 512 3 2 1" "4 128 128 128 256 3 2 1" "8 256 64 64 512 3 2 1" "4 512 32 32 512 3 2 1" "4 1024 16 16 512 3 1 1" "8 128 32 32 256 5 1 2" "8 256 16 16 256 5 1 2")
for g in "${G[@]}"; do
  echo "== $g"
  python tools/conv_bench.py $g --kinds fwd,dgrad_t --reps 20 --dtype bf16
  for sp in 1 2 4; do
    echo "splits $sp"
    GHM_LP_SPLITS=$sp python tools/conv_bench.py $g --kinds fwd,dgrad_t --reps 20 --dtype bf16
  done
done
