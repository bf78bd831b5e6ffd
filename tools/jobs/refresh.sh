cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r05 > gpurun_out/refresh_log.txt 2>&1
tail -5 gpurun_out/refresh_log.txt
python -c "
import json
d=json.loads(open('gpurun_out/refresh/r05_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['dtype'], d.get('steady_state'), d.get('value_fp32_mfma'))
for s in d.get('secondary',[]): print(s.get('name'), s.get('value'), s.get('ms_per_step'), s.get('error'))
print(d['roofline'])
"
