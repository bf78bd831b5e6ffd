cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_step.py -q -m gpu --timeout 600 2>&1 | tail -5
for g in "8 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1" "4 256 64 64 512 3 2 1"; do
  echo "== $g"
  python tools/conv_bench.py $g --kinds dgrad_t --reps 30 --dtype split --q q | tr '\n' '|'; echo
  python tools/conv_bench.py $g --kinds dgrad_t --reps 30 --dtype split2 --q q | tr '\n' '|'; echo
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16x3', d['value'], d['ms_per_step'])"
for g in "4 1024 64 64 256 3 1 1" "4 1024 32 32 512 3 1 1" "4 512 128 128 128 3 1 1"; do
  echo "== $g"
  python tools/conv_bench.py $g --kinds fwd,dgrad_t --reps 30 --dtype split --q q | tr '\n' '|'; echo
  GHM_SPLIT_BM128=1 python tools/conv_bench.py $g --kinds fwd,dgrad_t --reps 30 --dtype split --q q | tr '\n' '|'; echo
  GHM_SPLIT_BM128=1 GHM_SPLIT_SPLITS=2 python tools/conv_bench.py $g --kinds fwd,dgrad_t --reps 30 --dtype split --q q | tr '\n' '|'; echo
done
bash tools/jobs/thin.sh
