cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for env in "X=1" "GHM_GRAD_STREAM_PER_STAGE=1"; do
echo -n "$env bench: "; env $env timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('steady_state')['value'])"
done; done > gpurun_out/g19.txt
sort gpurun_out/g19.txt
