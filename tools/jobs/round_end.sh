cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|AssertionError|Error" | head -40 > gpurun_out/gputests_full.txt
cat gpurun_out/gputests_full.txt
bash tools/refresh_profiles.sh r05 > gpurun_out/refresh_log.txt 2>&1
bash tools/pmc_mfma.sh r05 "bf16x3" > gpurun_out/pmc_log.txt 2>&1
python -c "
import json
d=json.loads(open('gpurun_out/refresh/r05_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['dtype'], d.get('steady_state'), d.get('value_fp32_mfma'))
for s in d.get('secondary',[]): print(s.get('name'), s.get('value'), s.get('ms_per_step'), s.get('error'))
print(d['roofline'])
"
