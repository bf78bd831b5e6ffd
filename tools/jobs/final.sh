cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_lp.py tests/test_gpu_split.py tests/test_gpu_fullsize.py -q -m gpu --timeout 900 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|AssertionError|Error" | head -40 > gpurun_out/gputests.txt
cat gpurun_out/gputests.txt
for g in "8 128 128 128 128 5 1 2" "4 256 128 128 128 3 1 1" "8 64 256 256 128 3 2 1"; do
  echo -n "$g bf16 q: "; timeout 120 python tools/conv_bench.py $g --kinds wgrad --reps 30 --dtype bf16 --q q | awk '{printf "%s %s %s | ", $1, $3, $5}'; echo
done
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'])"
bash tools/refresh_profiles.sh r05 > gpurun_out/refresh_log.txt 2>&1
python -c "
import json
d=json.loads(open('gpurun_out/refresh/r05_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['dtype'], d.get('steady_state'), d.get('value_fp32_mfma'))
for s in d.get('secondary',[]): print(s.get('name'), s.get('value'), s.get('ms_per_step'), s.get('error'))
print(d['roofline'])
"
