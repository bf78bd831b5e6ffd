cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -3 > gpurun_out/g16.txt
for g in "8 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1" "8 256 64 64 512 3 2 1" "4 64 256 256 128 3 2 1"; do
  for env in "X=1" "GHM_SPLIT_DGRAD_S2_V1=1"; do
    echo -n "$g $env: "; env $env timeout 120 python tools/conv_bench.py $g --kinds dgrad_t --reps 30 --dtype split --q q | awk '{printf "%s %s %s | ", $1, $3, $5}'
    env $env timeout 120 python tools/conv_bench.py $g --kinds dgrad_t --reps 30 --dtype split2 --q q | awk '{printf "x2 %s %s | ", $3, $5}'; echo
  done
done >> gpurun_out/g16.txt 2>&1
for env in "X=1" "GHM_SPLIT_DGRAD_S2_V1=1" "X=2" "GHM_SPLIT_DGRAD_S2_V1=1"; do
echo -n "$env bench: "; env $env timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
done >> gpurun_out/g16.txt
cat gpurun_out/g16.txt
