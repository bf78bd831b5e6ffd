run() { python bench.py --steps 6 --warmup 2 --dtype bf16x3 --no-cpu-baseline --no-secondary "$@" 2>/tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['losses'][:2])" || tail -5 /tmp/err.txt; }
echo -n "dcgan   "; run --mode dcgan
echo -n "p2p     "; run --mode p2p
echo -n "config1 "; run --config1
echo -n "1024    "; run --in-shp 1024 --batch-per-gpu 2
echo -n "b8      "; run --batch-per-gpu 8
echo -n "graph   "; run --graph
echo -n "eager   "; run --issue eager
echo -n "1stream "; run --one-stream
