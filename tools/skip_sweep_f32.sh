#!/bin/bash
# in-step worth of the fp32 step's kernel classes (method of tools/skip_sweep.sh)
for k in "" "conv_patch_kernel" "(wgrad_patch_kernel" "dgrad_s2_patch_kernel" "conv_patch_kernel,(wgrad_patch_kernel,dgrad_s2_patch_kernel" \
         "fanout_kernel,fanin_s1_kernel,fanin_s2_kernel,thin_wgrad_kernel,taps_as_rows,pool_thin" "igemm_kernel,(wgrad_kernel,direct_smallr,smallk_dgrad,igemm_splitk,reduce_splits" \
         "bn_" "maxpool" "transpose,collapse,expand" "up_bilinear,pp_to_hi,hi_to_pp" "channel_sum,act_bwd" "rmsprop,adam"; do
  echo -n "skip ${k:-nothing}: "
  env ${k:+GHM_SKIP_KERNELS="$k"} python bench.py --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
