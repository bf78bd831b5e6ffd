#!/bin/bash
# Regenerates the files under profiles/ for the current round (run through gpurun from the repo root, then copy
# gpurun_out/refresh/* into profiles/):   gpurun --timeout 2400 -- 'tools/refresh_profiles.sh r04'
R=${1:-r04}
O=gpurun_out/refresh
mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/${R}_bench.json 2> $O/${R}_bench.err          # the driver's command: headline + secondary lines
python bench.py --steps 30 --warmup 5 --dtype bf16 --no-cpu-baseline --no-secondary > $O/${R}_bench_bf16.json 2>> $O/${R}_bench.err
python bench.py --steps 30 --warmup 5 --dtype f16 --no-cpu-baseline --no-secondary > $O/${R}_bench_f16.json 2>> $O/${R}_bench.err
python bench.py --steps 30 --warmup 5 --dtype bf16x3 --no-cpu-baseline --no-secondary > $O/${R}_bench_bf16x3.json 2>> $O/${R}_bench.err
python bench.py --steps 10 --warmup 3 --in-shp 1024 --no-cpu-baseline --no-secondary > $O/${R}_bench_1024_f32.json 2>> $O/${R}_bench.err
python bench.py --steps 10 --warmup 3 --in-shp 1024 --dtype f16 --no-cpu-baseline --no-secondary > $O/${R}_bench_1024_f16.json 2>> $O/${R}_bench.err
python bench.py --steps 20 --warmup 3 --batch-per-gpu 8 --no-cpu-baseline --no-secondary > $O/${R}_bench_b8.json 2>> $O/${R}_bench.err
python bench.py --steps 20 --warmup 3 --graph --no-cpu-baseline --no-secondary > $O/${R}_bench_graph.json 2>> $O/${R}_bench.err
GHM_PROFILE_ALL=1 python bench.py --steps 5 --profile --no-cpu-baseline --no-secondary > /dev/null 2> $O/${R}_kernel_table.txt
GHM_PROFILE_ALL=1 python bench.py --steps 5 --profile --no-cpu-baseline --no-secondary --dtype bf16 > /dev/null 2> $O/${R}_kernel_table_bf16.txt
GHM_PROFILE_ALL=1 python bench.py --steps 5 --profile --no-cpu-baseline --no-secondary --dtype bf16x3 > /dev/null 2> $O/${R}_kernel_table_bf16x3.txt
( [ -x tools/mfma_bf16_probe ] || hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_probe.hip -o tools/mfma_bf16_probe ) 2>/dev/null; tools/mfma_bf16_probe > $O/${R}_mfma_probe.txt 2>&1
bash tools/split_quick.sh > $O/${R}_split_kernels_alone.txt 2>&1
for dt in f32 bf16 bf16x3; do
  rm -rf /tmp/prof_$dt
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$dt -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --dtype $dt > $GRAFT_REPO_ROOT/$O/${R}_bench_under_rocprof_$dt.json 2>/dev/null)
  f=$(find /tmp/prof_$dt -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${R}_rocprofv3_kernel_stats_$dt.csv
done
bash tools/upper_bounds.sh bf16 > $O/${R}_instep_skip_sweep_bf16.txt 2>&1
bash tools/upper_bounds.sh f32 > $O/${R}_instep_skip_sweep_f32.txt 2>&1
python tools/train_throughput.py 100 bf16 > $O/${R}_train_throughput.txt 2>&1
python tools/train_throughput.py 60 f32 >> $O/${R}_train_throughput.txt 2>&1
python tools/train_throughput.py 100 bf16x3 >> $O/${R}_train_throughput.txt 2>&1
ls -la $O
