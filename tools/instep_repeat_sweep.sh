#!/bin/bash
# the joint step with one kernel CLASS issued twice (bench.py --repeat; timing only): the increase is what the class costs
# inside the overlapped three-stream schedule ON REAL DATA (tools/split_skip_sweep.sh over-states producers: skipped, they leave
# zeros in matrix-core operands, which run ~18 % faster).  Columns: class, ms per step, increase over the plain step.
DT=${1:-bf16x3}
run() { python bench.py --dtype $DT --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --minimal "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
b1=$(run); b2=$(run)
echo "plain: $b1 $b2"
for k in \
  "sp_conv2_kernel<5" "sp_conv2_kernel<3" "sp_wgrad_kernel" "sp_dgrad_s2_kernel" \
  "fanout_kernel,thin_wgrad_kernel,pool_thin,fanin_s1_kernel,fanin_s2_kernel,taps_as_rows" \
  "igemm_kernel,wgrad_kernel,direct_smallr,smallk_dgrad,dense" \
  "bn_fwd" "bn_bwd" "maxpool_mask_bwd,maxpool_bwd,maxpool_fwd" "up_bilinear_fwd,up_bilinear_bwd" "q_pack,lp_pack" "bias_grad,act_bwd"; do
  t=$(run --repeat "$k")
  python -c "print('%-90s %7.3f  +%.3f' % ('$k', $t, $t - ($b1 + $b2) / 2))"
done
