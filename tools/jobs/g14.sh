cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for env in "X=1" "GHM_SPLIT_WGRAD_ROUNDS=2" "GHM_SPLIT_WGRAD_ROUNDS=3" "GHM_SPLIT_WGRAD_ROUNDS=4" "GHM_SPLIT_WGRAD_ROUNDS=0.5" "GHM_SIDE_NETS=GDPU" "GHM_SIDE_NETS=PU"; do
    echo -n "$env: "; env $env timeout 300 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
done; done > gpurun_out/g14.txt 2>&1
cat gpurun_out/g14.txt
