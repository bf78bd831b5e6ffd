run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['losses'])"; }
for dt in f32 bf16x3 bf16; do
  echo "== $dt recorded x2, eager, one-stream"
  run --dtype $dt; run --dtype $dt; run --dtype $dt --issue eager; run --dtype $dt --one-stream
done
