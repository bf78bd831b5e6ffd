cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mul in 1 4 16; do
  rm -rf $R/gpurun_out/kstat
  GHM_THIN_BLOCKS_MUL=$mul rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstat -o s -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/kstat_bench.json 2>/dev/null
  MUL=$mul python - <<'PY'
import csv, json, os
R=os.environ['GRAFT_REPO_ROOT']
d=json.loads(open(R+'/gpurun_out/kstat_bench.json').read().strip().splitlines()[-1])
print("== mul %s: %.1f img/s"%(os.environ['MUL'], d['value']))
for r in csv.DictReader(open(R+'/gpurun_out/kstat/s_kernel_stats.csv')):
    if any(k in r['Name'] for k in ('fanout','thin_wgrad','fanin')):
        print("   %-52s calls/step %5.1f avg %8.0f us"%(r['Name'][5:57], int(r['Calls'])/23, float(r['AverageNs'])/1e3))
PY
done
