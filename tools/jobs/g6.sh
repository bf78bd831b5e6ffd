cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -6 > gpurun_out/g6_tests.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "reduced_precision" 2>&1 | grep -E "full-size batch-4|passed|failed|Error|assert" >> gpurun_out/g6_tests.txt
timeout 600 python -m pytest tests/test_gpu_f4.py -x -q -k "instance" 2>&1 | tail -4 >> gpurun_out/g6_tests.txt
for g in "8 128 128 128 128 5 1 2" "4 256 128 128 128 3 1 1" "8 64 256 256 128 3 2 1"; do
  echo "== $g"
  for dt in split split2 bf16; do
    echo -n "$dt: "; timeout 120 python tools/conv_bench.py $g --kinds fwd,dgrad_t,wgrad --reps 30 --dtype $dt --q q | awk '{printf "%s %s %s | ", $1, $3, $5}'; echo
  done
done > gpurun_out/g6_quick.txt 2>&1
for dt in bf16x2 bf16x3 bf16; do
timeout 300 python bench.py --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/g6_bench_$dt.json 2> gpurun_out/g6_bench_$dt.err
done
cat gpurun_out/g6_tests.txt gpurun_out/g6_quick.txt
python - <<'PY'
import json
for dt in ['bf16x2','bf16x3','bf16']:
    f='gpurun_out/g6_bench_%s.json'%dt
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('steady_state'), d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['frac_isolated'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
