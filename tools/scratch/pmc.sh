cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GEOM="${GEOM:-4 512 128 128 128 3 1 1}"
KIND="${KIND:-wgrad}"
PAT="${PAT:-wgrad_patch}"
rm -rf $R/gpurun_out/pmc_k
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_k/p$i -- python $R/tools/conv_bench.py $GEOM --kinds $KIND --reps 3 > /dev/null 2>&1
done
PAT=$PAT python - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']; pat=os.environ['PAT']
acc=collections.defaultdict(lambda: [0.0,0])
for f in glob.glob(R+'/gpurun_out/pmc_k/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
for k,(v,n) in sorted(acc.items()): print("%-34s %16.0f  (per dispatch, %d samples)"%(k, v/n, n))
PY
