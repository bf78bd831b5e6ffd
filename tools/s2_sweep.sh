# tuning sweep used in round 2 (isolated, warm rates through tools/conv_bench.py): stride-2 family and weight gradients;
# GHM_ABLATE=1 skips the staging loads, =2 the LDS stores too; GHM_[LP_]DGRAD_S2_TILE / _SPLITS, GHM_LP_SPLITS,
# GHM_WGRAD_BKP16 override the host-side plans
G=("8 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1" "4 256 64 64 512 3 2 1" "4 128 128 128 256 3 2 1"
   "4 512 128 128 128 3 1 1" "4 1024 64 64 256 3 1 1" "4 256 256 256 64 3 1 1" "8 64 256 256 128 5 1 2" "8 128 128 128 128 5 1 2")
for g in "${G[@]}"; do
  python tools/conv_bench.py $g --kinds fwd,dgrad_t,wgrad
  python tools/conv_bench.py $g --kinds fwd,dgrad_t,wgrad --dtype bf16
done
