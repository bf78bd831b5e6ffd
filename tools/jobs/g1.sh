cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -5 > gpurun_out/g1_tests.txt
for g in "8 128 128 128 128 5 1 2" "8 64 256 256 64 5 1 2" "4 256 128 128 128 3 1 1" "4 128 256 256 64 3 1 1" "8 64 256 256 128 3 2 1" "4 1024 64 64 256 3 1 1"; do
  echo "== $g"
  timeout 120 python tools/conv_bench.py $g --kinds fwd,dgrad_t --reps 30 --dtype split --q q | tr '\n' '|'; echo
  GHM_SPLIT_V1=1 timeout 120 python tools/conv_bench.py $g --kinds fwd,dgrad_t --reps 30 --dtype split --q q | tr '\n' '|'; echo
done > gpurun_out/g1_quick.txt 2>&1
timeout 300 python bench.py --dtype bf16x3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/g1_bench_x3.json 2> gpurun_out/g1_bench_x3.err
GHM_SPLIT_V1=1 timeout 300 python bench.py --dtype bf16x3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/g1_bench_x3_v1.json 2>> gpurun_out/g1_bench_x3.err
cat gpurun_out/g1_tests.txt gpurun_out/g1_quick.txt
python - <<'PY'
import json
for f in ['gpurun_out/g1_bench_x3.json','gpurun_out/g1_bench_x3_v1.json']:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
