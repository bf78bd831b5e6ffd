cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|AssertionError|Error" | head -60 > gpurun_out/gputests_full.txt
cat gpurun_out/gputests_full.txt
timeout 300 python -m pytest tests/test_gpu_split.py -q -m gpu -k test_split_train_step_meets_the_fp32_bounds -s 2>&1 | grep -E "^step|passed|failed" | head -16
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16x3', d['value'], d['ms_per_step'])"
