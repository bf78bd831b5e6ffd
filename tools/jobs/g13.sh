cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -3 > gpurun_out/g13.txt
for g in "8 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1" "8 256 64 64 512 3 2 1" "4 512 32 32 512 3 2 1"; do
  echo "== $g"
  for dt in split split2; do
    echo -n "$dt: "; timeout 120 python tools/conv_bench.py $g --kinds fwd,dgrad_t,wgrad --reps 30 --dtype $dt --q q | awk '{printf "%s %s %s | ", $1, $3, $5}'; echo
  done
done >> gpurun_out/g13.txt 2>&1
timeout 300 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d.get('steady_state'))" >> gpurun_out/g13.txt
cat gpurun_out/g13.txt
