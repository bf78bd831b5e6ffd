cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_split.py -q -m gpu --timeout 600 -k "dgrad or s2 or dact" 2>&1 | tail -3
GHM_PROFILE_ALL=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --profile 2> gpurun_out/prof_all.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16x3', d['value'], d['ms_per_step'])"
grep -E "sp_dgrad_s2" gpurun_out/prof_all.txt | head -20
