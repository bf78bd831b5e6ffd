cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -3 > gpurun_out/g17.txt
for g in "8 128 128 128 128 5 1 2" "8 64 256 256 128 5 1 2" "4 128 256 256 64 3 1 1" "8 64 256 256 128 3 2 1"; do
  for env in "GHM_SPLIT_PERSIST=1" "GHM_SPLIT_PERSIST=0" "GHM_SPLIT_PERSIST=0.5"; do
    echo -n "$g $env: "; env $env timeout 120 python tools/conv_bench.py $g --kinds fwd,dgrad_t --reps 30 --dtype split --q q | awk '{printf "%s %s %s | ", $1, $3, $5}'; echo
  done
done >> gpurun_out/g17.txt 2>&1
for rep in 1 2; do for env in "GHM_SPLIT_PERSIST=1" "GHM_SPLIT_PERSIST=0" "GHM_SPLIT_PERSIST=0.5"; do
echo -n "$env bench: "; env $env timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
done; done >> gpurun_out/g17.txt
cat gpurun_out/g17.txt
