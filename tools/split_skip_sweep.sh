#!/bin/bash
# the split-fp32 step with one kernel CLASS skipped (GHM_SKIP_KERNELS; results wrong, timing only): what each class is worth
# in the overlapped schedule.  CAVEAT (measured): skipping a kernel that PRODUCES matrix-core operands (sp_pack_*, the packs) leaves
# zeros in them, and the bf16 MFMA kernels run ~18 % faster on zero operands (power management): those two lines
# over-state their class several times (1.36 / 0.92 ms in the sweep against 0.18 / 0.28 ms of kernels alone)
run() { env $1 python bench.py --dtype bf16x3 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for k in "X=1" "X=2" \
  "GHM_SKIP_KERNELS=sp_conv2_kernel" \
  "GHM_SKIP_KERNELS=sp_wgrad_kernel" \
  "GHM_SKIP_KERNELS=sp_dgrad_s2_kernel" \
  "GHM_SKIP_KERNELS=sp_pack_kernel" \
  "GHM_SKIP_KERNELS=sp_pack_w" \
  "GHM_SKIP_KERNELS=fanout_kernel,thin_wgrad_kernel,pool_thin,fanin_s1_kernel,fanin_s2_kernel,taps_as_rows,shift_,direct_smallr,smallk_dgrad" \
  "GHM_SKIP_KERNELS=igemm_kernel,(wgrad_kernel,igemm_splitk,dense_smallp" \
  "GHM_SKIP_KERNELS=bn_" \
  "GHM_SKIP_KERNELS=maxpool2_mask_bwd,maxpool2_" \
  "GHM_SKIP_KERNELS=up_bilinear" \
  "GHM_SKIP_KERNELS=reduce_splits_wide_kernel,reduce_splits_kernel,channel_sum" \
  "GHM_SKIP_KERNELS=rmsprop,transpose_weights,upconv_collapse,upconv_expand"; do
  echo -n "$k: "; run "$k"
done
