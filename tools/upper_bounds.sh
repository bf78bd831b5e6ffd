#!/bin/bash
# UPPER BOUNDS of fusions not built: the step with the launches a fusion would remove skipped (GHM_SKIP_KERNELS; results are
# wrong, timing only).  step(all) - step(without) = the most the fusion could gain in the overlapped schedule.
dt=${1:-bf16}
run() { env $1 python bench.py --dtype $dt --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for k in "X=1" "X=2" \
  "GHM_SKIP_KERNELS=bn_rows_partial<false>,bn_stats_final,bn_stats_partial" \
  "GHM_SKIP_KERNELS=up_bilinear_fwd" \
  "GHM_SKIP_KERNELS=up_bilinear_bwd" \
  "GHM_SKIP_KERNELS=lp_pack_batched_kernel" \
  "GHM_SKIP_KERNELS=reduce_splits_wide_kernel,reduce_splits_kernel" \
  "GHM_SKIP_KERNELS=(wgrad_kernel" \
  "GHM_SKIP_KERNELS=channel_sum" \
  "GHM_SKIP_KERNELS=q_pack_kernel" \
  "GHM_SKIP_KERNELS=maxpool2_mask_bwd" \
  "GHM_SKIP_KERNELS=pool_thin" \
  "GHM_SKIP_KERNELS=fanout_kernel" \
  "GHM_SKIP_KERNELS=thin_wgrad_kernel" \
  "GHM_SKIP_KERNELS=fanin_s1_kernel,fanin_s2_kernel,taps_as_rows,shift_" \
  "GHM_SKIP_KERNELS=rmsprop" \
  "GHM_SKIP_KERNELS=bn_bwd,bn_rows_partial<true>" \
  "GHM_SKIP_KERNELS=bn_apply"; do
  echo -n "$k: "; run "$k"
done
