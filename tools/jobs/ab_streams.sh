export TMPDIR=/tmp
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary"
r() { echo "== $1"; shift; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
r default X=1
r side_PU GHM_SIDE_NETS=PU
r side_GDPU GHM_SIDE_NETS=GDPU
r per_stage GHM_GRAD_STREAM_PER_STAGE=1
r persist128 GHM_SPLIT_PERSIST=0.5
done
rm -rf /tmp/tl_x3
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_x3 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1)
python tools/timeline.py /tmp/tl_x3 > gpurun_out/s3/timeline.txt 2>&1
