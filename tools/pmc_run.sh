#!/bin/bash
# PMC passes for profiles/: counter calibration, then FETCH_SIZE / WRITE_SIZE per kernel of the f32 and bf16 steps
# (counters only: --pmc with --kernel-trace, nothing else).  NOTE: these passes run the step with --one-stream --issue eager, not
# the three-stream recorded step bench.py times (rocprofv3 serialises dispatches under --pmc either way); tools/pmc_mfma.sh
# re-takes FETCH / WRITE in the bench's own form (profiles/r04_pmc_traffic_3stream_*.json).   gpurun --timeout 1500 -- 'tools/pmc_run.sh r03'
R=${1:-r03}
O=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/pmc_calibrate.hip -o /tmp/pmc_cal 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -- /tmp/pmc_cal > /dev/null 2>&1
  f=$(find /tmp/cal_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c >> $O/${R}_pmc_calibration.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-10s %-12s counter*1024 / bytes moved = %.3f" % (sys.argv[2], k, sum(v) / len(v) * 1024 / (4 * 2**28)))
PY
done
cat $O/${R}_pmc_calibration.txt
for dt in f32 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${dt}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${dt}_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --one-stream --issue eager --dtype $dt > /dev/null 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pmc_${dt}_FETCH_SIZE /tmp/pmc_${dt}_WRITE_SIZE > $O/${R}_pmc_traffic_$dt.json
done
ls -la $O
