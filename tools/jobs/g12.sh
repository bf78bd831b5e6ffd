cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for env in "X=1" "GHM_SKIP_STREAM_WAITS=1" "X=2" "GHM_SKIP_STREAM_WAITS=1"; do
    echo -n "$env: "; env $env timeout 300 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
done > gpurun_out/g12.txt 2>&1
echo -n "no-grad-streams: " >> gpurun_out/g12.txt; timeout 300 python bench.py --dtype bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-grad-streams 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('steady_state'))" >> gpurun_out/g12.txt
rm -rf gpurun_out/tl
GHM_SKIP_STREAM_WAITS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --dtype bf16x3 --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2> gpurun_out/g12.err
python tools/timeline.py gpurun_out/tl > gpurun_out/g12_timeline_nowaits.txt 2>&1
rm -rf gpurun_out/tl
cat gpurun_out/g12.txt; head -40 gpurun_out/g12_timeline_nowaits.txt
