cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in dgrad_t fwd wgrad; do
 for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM"; do
  rm -rf /tmp/p1
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/p1 -- python $R/tools/conv_bench.py 8 64 256 256 128 3 2 1 --kinds $kind --reps 10 --warm-ms 20 --dtype split --q q > /dev/null 2>&1
  f=$(find /tmp/p1 -name "*counter_collection.csv" | head -1)
  python - "$f" "$kind" <<'PY'
import csv,sys,collections
f,kind=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    if 'sp_' not in k or 'pack' in k: continue
    acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
for k,v in acc.items():
    print(kind, k[:60], {c: '%.3g'%x for c,x in v.items()})
PY
 done
done
