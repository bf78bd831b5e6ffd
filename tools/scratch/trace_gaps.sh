cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ktrace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktrace -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(R+'/gpurun_out/ktrace/**/*kernel_trace.csv', recursive=True)[0]
rows=[(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# take the last 40% of the trace (timed region steady state)
t0=rows[int(len(rows)*0.6)][0]; t1=max(r[1] for r in rows)
sel=[r for r in rows if r[0]>=t0]
ev=[]
for s,e,_ in sel: ev.append((s,1)); ev.append((e,-1))
ev.sort()
busy=0; conc_time={}; cur=0; last=t0
for t,d in ev:
    if t>last:
        conc_time[cur]=conc_time.get(cur,0)+(t-last)
        last=t
    cur+=d
tot=t1-t0
print("window %.2f ms, kernels %d"%(tot/1e6, len(sel)))
for k in sorted(conc_time): print("  concurrency %d: %.1f%%"%(k, 100*conc_time[k]/tot))
# biggest idle gaps
PY
