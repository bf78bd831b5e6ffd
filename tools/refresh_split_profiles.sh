R=${1:-r05}; O=gpurun_out/refresh; mkdir -p $O; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/${R}_bench.json 2> $O/${R}_bench.err
python bench.py --steps 30 --warmup 5 --dtype bf16x3 --no-cpu-baseline --no-secondary > $O/${R}_bench_bf16x3.json 2>> $O/${R}_bench.err
GHM_PROFILE_ALL=1 python bench.py --steps 5 --profile --no-cpu-baseline --no-secondary --dtype bf16x3 > /dev/null 2> $O/${R}_kernel_table_bf16x3.txt
rm -rf /tmp/prof_x3
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --dtype bf16x3 > $GRAFT_REPO_ROOT/$O/${R}_bench_under_rocprof_bf16x3.json 2>/dev/null)
f=$(find /tmp/prof_x3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${R}_rocprofv3_kernel_stats_bf16x3.csv
python tools/program_dump.py --dtype bf16x3 > $O/${R}_program_bf16x3.txt 2>&1
python tools/train_throughput.py 100 bf16x3 > $O/${R}_train_throughput_bf16x3.txt 2>&1
