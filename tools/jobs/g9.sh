cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --dtype bf16x3 --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/g9_bench.json 2> gpurun_out/g9.err
python tools/timeline.py gpurun_out/tl --dump > gpurun_out/g9_timeline.txt 2>&1
rm -rf gpurun_out/tl
head -60 gpurun_out/g9_timeline.txt
