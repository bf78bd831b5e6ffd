#!/bin/bash
# MFMA-utilisation counters per kernel of the THREE-STREAM RECORDED step (the form bench.py times), f32 and bf16, and
# FETCH / WRITE traffic re-taken in the same form.  Counters only (--pmc with --kernel-trace).
#   gpurun --timeout 1500 -- 'tools/pmc_mfma.sh r04'   then copy gpurun_out/pmc/* into profiles/
R=${1:-r06}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --minimal"
for dt in ${2:-bf16x3 bf16x2 f32 bf16}; do
  rm -rf /tmp/mf_$dt /tmp/mf2_$dt
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
      --output-format csv -d /tmp/mf_$dt -- $B --dtype $dt > $O/${R}_pmc_mfma_$dt.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_mfma.py /tmp/mf_$dt > $O/${R}_pmc_mfma_$dt.json 2>> $O/${R}_pmc_mfma_$dt.log
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc3_${dt}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc3_${dt}_$c -- $B --dtype $dt > /dev/null 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pmc3_${dt}_FETCH_SIZE /tmp/pmc3_${dt}_WRITE_SIZE > $O/${R}_pmc_traffic_3stream_$dt.json
done
ls -la $O
