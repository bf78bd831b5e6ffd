cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -3 > gpurun_out/g4_tests.txt
for g in "8 128 128 128 128 5 1 2" "8 64 256 256 128 5 1 2" "4 256 128 128 128 3 1 1" "4 256 256 256 64 3 1 1" "4 1024 64 64 256 3 1 1" "8 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1" "4 1024 32 32 512 3 1 1"; do
  echo "== $g"
  for env in "X=1" "GHM_SPLIT_WGRAD_V1=1"; do
    echo -n "$env: "; env $env timeout 120 python tools/conv_bench.py $g --kinds wgrad --reps 30 --dtype split --q q | awk '{printf "%s %s %s | ", $1, $3, $5}'; echo
  done
done > gpurun_out/g4.txt 2>&1
timeout 300 python bench.py --dtype bf16x3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/g4_bench_x3.json 2> gpurun_out/g4_bench_x3.err
cat gpurun_out/g4_tests.txt gpurun_out/g4.txt
python - <<'PY'
import json
for f in ['gpurun_out/g4_bench_x3.json']:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
