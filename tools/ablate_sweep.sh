for dt in f32 bf16; do
 for a in "" "maxpool_fwd,maxpool_bwd,maxpool_mask_bwd" "bn_fwd" "bn_fwd,bn_bwd" "act_bwd" "bias_grad" "igemm_kernel,wgrad_kernel,direct_smallr,smallk_dgrad" "fanout_kernel,fanin_s1,fanin_s2,thin_wgrad,taps_as_rows" "lp_pack,lp_pack_t,transpose_w,collapse_w,expand_wgrad" "up_bilinear_fwd,up_bilinear_bwd,pp_to_hi,hi_to_pp" "conv_wgrad,upconv_wgrad,deconv_wgrad,dense_wgrad" "rmsprop_dcgan_gen,rmsprop_dcgan_disc,rmsprop_p2p_gen,rmsprop_p2p_disc"; do
  python bench.py --dtype $dt --no-cpu-baseline --steps 15 --ablate "$a" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', '%-60s' % '$a', d['ms_per_step'])"
 done
done
