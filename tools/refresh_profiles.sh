#!/bin/bash
# Regenerates the files under profiles/ for the current round (run through gpurun from the repo root, then copy
# gpurun_out/refresh/* into profiles/):   gpurun --timeout 2400 -- 'tools/refresh_profiles.sh r05'
# Round 5: the headline arithmetic is bf16x3 (python bench.py with no --dtype); f32 = the same step on v_mfma_f32_32x32x2_f32.
R=${1:-r06}
O=gpurun_out/refresh
mkdir -p $O
export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench.json 2> $O/${R}_bench.err          # the driver's command: headline + secondary lines
cp gpurun_out/bench_secondary.json $O/${R}_bench_secondary.json                                     # (its side file: the later runs overwrite it)
for dt in f32 bf16 bf16x2 f16; do
  python bench.py --steps 30 --warmup 5 --dtype $dt --no-cpu-baseline --no-secondary > $O/${R}_bench_$dt.json 2>> $O/${R}_bench.err
done
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-secondary > $O/${R}_bench_bf16x3_100steps.json 2>> $O/${R}_bench.err
python bench.py --steps 10 --warmup 3 --in-shp 1024 --no-cpu-baseline --no-secondary > $O/${R}_bench_1024_bf16x3.json 2>> $O/${R}_bench.err
python bench.py --steps 10 --warmup 3 --in-shp 1024 --dtype f16 --batch-per-gpu 2 --no-cpu-baseline --no-secondary > $O/${R}_bench_1024_f16_b2.json 2>> $O/${R}_bench.err
python bench.py --steps 20 --warmup 3 --batch-per-gpu 8 --no-cpu-baseline --no-secondary > $O/${R}_bench_b8.json 2>> $O/${R}_bench.err
for dt in bf16x3 bf16x2 f32; do
  GHM_PROFILE_ALL=1 python bench.py --steps 5 --profile --no-cpu-baseline --no-secondary --dtype $dt > /dev/null 2> $O/${R}_kernel_table_$dt.txt
done
bash tools/split_quick.sh > $O/${R}_split_kernels_alone.txt 2>&1
for dt in bf16x3 bf16x2 f32; do
  rm -rf /tmp/prof_$dt
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$dt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --dtype $dt > $GRAFT_REPO_ROOT/$O/${R}_bench_under_rocprof_$dt.json 2>/dev/null)
  f=$(find /tmp/prof_$dt -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${R}_rocprofv3_kernel_stats_$dt.csv
done
rm -rf /tmp/tl_x3
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1)
python tools/timeline.py /tmp/tl_x3 > $O/${R}_timeline_bf16x3.txt 2>&1
python tools/program_dump.py --dtype bf16x3 > $O/${R}_program_bf16x3.txt 2>&1
python tools/train_throughput.py 100 bf16x3 > $O/${R}_train_throughput.txt 2>&1
python tools/train_throughput.py 100 bf16x2 >> $O/${R}_train_throughput.txt 2>&1
python tools/mode_agreement.py 2 9 10 11 12 13 14 > $O/${R}_mode_agreement.txt 2>&1
python tools/step_time_series.py --windows 30 > $O/${R}_step_time_series.txt 2>&1
python tools/host_issue_time.py bf16x3 > $O/${R}_host_issue_time.txt 2>&1
ls -la $O
