for q in 2 3 4 6 8 12 16 24; do
echo -n "GPU_MAX_HW_QUEUES=$q: "
for dt in f32 bf16; do GPU_MAX_HW_QUEUES=$q python bench.py --steps 30 --warmup 5 --dtype $dt --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['dtype'], d['value'], end='   ')"; done; echo
done
