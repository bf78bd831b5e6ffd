cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for env in "GHM_SPLIT_PERSIST=0" "GHM_SPLIT_PERSIST=0.5" "GHM_SPLIT_PERSIST=0.75" "GHM_SPLIT_PERSIST=0.375" "GHM_SPLIT_PERSIST=0.625"; do
echo -n "$env bench: "; env $env timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('steady_state')['value'])"
done; done > gpurun_out/g18.txt
sort gpurun_out/g18.txt
