export TMPDIR=/tmp
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary"
r() { echo "== $1"; shift; env "$@" $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
r default X=1
r persist0 GHM_SPLIT_PERSIST=0
r persist0.97 GHM_SPLIT_PERSIST=0.97
r persist0.94 GHM_SPLIT_PERSIST=0.94
r persist0.875 GHM_SPLIT_PERSIST=0.875
r bm128 GHM_SPLIT_BM128=1
done
