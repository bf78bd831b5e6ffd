run() { python bench.py --steps 20 --warmup 5 --dtype bf16x3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo -n "base            "; run
echo -n "BM64            "; GHM_SPLIT_BM64=1 run
echo -n "wgrad rounds 2  "; GHM_SPLIT_WGRAD_ROUNDS=2 run
echo -n "wgrad rounds .5 "; GHM_SPLIT_WGRAD_ROUNDS=0.5 run
echo -n "side PU         "; GHM_SIDE_NETS=PU run
echo -n "side DPUG       "; GHM_SIDE_NETS=DPUG run
echo -n "no narrow       "; GHM_SPLIT_NO_NARROW=1 run
echo -n "base            "; run
