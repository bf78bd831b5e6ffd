#!/bin/bash
# finer classes than tools/skip_sweep.sh (same method)
dt=${1:-bf16}
for k in "" "reduce_splits_kernel" "igemm_splitk_epilogue" "q_pack_kernel" "lp_pack_batched" "bn_rows_partial<false>,bn_stats_final" "bn_rows_partial<true>,bn_bwd_final" \
         "bn_apply,bn_bwd_apply" "bn_fwd_small,bn_bwd_small" "thin_wgrad_kernel" "fanout_kernel" "fanin_s1,fanin_s2" "maxpool2_mask_bwd" "igemm_kernel" "(wgrad_kernel" "wgrad_patch_kernel" \
         "smallk_dgrad,direct_smallr" "up_bilinear" "channel_sum"; do
  echo -n "skip ${k:-nothing}: "
  env ${k:+GHM_SKIP_KERNELS="$k"} python bench.py --dtype $dt --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
