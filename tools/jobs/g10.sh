cd $GRAFT_REPO_ROOT
for env in "X=1" "GHM_REC_EVENT_PER_REPLAY=1" "X=2" "GHM_REC_EVENT_PER_REPLAY=1"; do
  for dt in bf16x3 bf16; do
    echo -n "$env $dt: "; env $env timeout 300 python bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
  done
done > gpurun_out/g10.txt 2>&1
cat gpurun_out/g10.txt
