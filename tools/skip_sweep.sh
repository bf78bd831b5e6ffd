#!/bin/bash
# what each kernel class is worth IN THE STEP: the train step with the launches of one class skipped (GHM_SKIP_KERNELS,
# substring match on the kernel name; results are wrong, timing only).  step(all) - step(without class) = the class's
# share of the critical path, to hold against its isolated time in profiles/*kernel_table*.
dt=${1:-bf16}
for k in "" "lp_conv_kernel" "lp_wgrad_q_kernel" "lp_dgrad_s2_kernel" "lp_wgrad_kernel" "lp_conv_kernel,lp_wgrad_q_kernel,lp_dgrad_s2_kernel,lp_wgrad_kernel" \
         "fanout_kernel,fanin_s1_kernel,fanin_s2_kernel,thin_wgrad_kernel,taps_as_rows" "igemm_kernel,wgrad_kernel,direct_smallr,smallk_dgrad,wgrad_patch_kernel,igemm_splitk,reduce_splits" \
         "bn_" "maxpool" "q_pack,q_unpack,lp_pack,transpose,collapse,expand" "up_bilinear,pp_to_hi,hi_to_pp" "channel_sum,act_bwd,q_rows_sum" "rmsprop,adam"; do
  echo -n "skip ${k:-nothing}: "
  env ${k:+GHM_SKIP_KERNELS=$k} python bench.py --dtype $dt --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
