cd $GRAFT_REPO_ROOT
GHM_PROFILE_ALL=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --profile 2> gpurun_out/prof_all.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16x3', d['value'], d['ms_per_step'])"
grep -E "sp_dgrad_s2|sp_conv2_kernel<3, 2>|sp_conv2_kernel<3, 1>" gpurun_out/prof_all.txt | head -60
for g in "4 64 256 256 128 3 2 1" "4 128 128 128 256 3 2 1" "4 256 64 64 512 3 2 1" "8 256 64 64 512 3 2 1"; do
  echo "== $g"
  python tools/conv_bench.py $g --kinds fwd,dgrad_t,wgrad --reps 30 --dtype split --q q | tr '\n' '|'; echo
done
