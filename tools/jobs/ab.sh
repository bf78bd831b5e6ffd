cd $GRAFT_REPO_ROOT
b() { env $1 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary --minimal 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
b X=0; b GHM_SPLIT_BM128=1; b X=0; b GHM_SPLIT_BM128=1
