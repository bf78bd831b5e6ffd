#!/bin/bash
# where the waves of ONE kernel spend their cycles (SQ stall buckets, MI355X_MICROARCH.md "PMC slots"): run through gpurun
#   tools/pmc_stalls.sh "8 128 128 128 128 5 1 2" fwd split
G=${1:-"8 128 128 128 128 5 1 2"}; K=${2:-fwd}; DT=${3:-split}
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/stall && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d /tmp/stall -- python $GRAFT_REPO_ROOT/tools/conv_bench.py $G --kinds $K --reps 10 --warm-ms 5 --dtype $DT $( [ $DT != f32 ] && echo --q q ) > /dev/null 2>&1
python - <<'P'
import csv, glob, collections
f = glob.glob('/tmp/stall/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    if len(v.get('SQ_WAVE_CYCLES', [])) < 5: continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m['SQ_WAVE_CYCLES']
    print(k)
    print('   wave-cycles %.3g | parked (waitcnt/barrier) %.1f%% | issue-stall %.1f%% (of which LDS %.1f%%) | issuing %.1f%% | LDS busy: conflicts %.3g of %.3g array cycles | MFMA busy cycles %.3g'
          % (wc, 100 * m['SQ_WAIT_ANY'] / wc, 100 * m['SQ_WAIT_INST_ANY'] / wc, 100 * m['SQ_WAIT_INST_LDS'] / wc, 100 * m['SQ_ACTIVE_INST_ANY'] / wc,
             m['SQ_LDS_BANK_CONFLICT'], m['SQ_LDS_IDX_ACTIVE'], m['SQ_VALU_MFMA_BUSY_CYCLES']))
P
