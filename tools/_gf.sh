for dt in f32 bf16; do
for sw in "X=1" "GHM_EVENT_SYSTEM_SCOPE=1" "X=2" "GHM_EVENT_SYSTEM_SCOPE=1"; do
echo -n "$dt $sw: "; env $sw python bench.py --steps 30 --warmup 5 --dtype $dt --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['losses'][:2])"
done; done
