python -m pytest tests/test_gpu_step.py -x -q > gpurun_out/t.txt 2>&1; tail -3 gpurun_out/t.txt
python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain  ', d['value'], d['ms_per_step'], d['losses'])"
GHM_NO_CHAIN_STREAM=1 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nochain', d['value'], d['ms_per_step'], d['losses'])"
python bench.py --steps 20 --no-cpu-baseline --no-grad-streams 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain  ', d['value'], d['ms_per_step'], d['losses'])"
