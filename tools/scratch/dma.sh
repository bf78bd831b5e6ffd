python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | grep -E "passed|failed|Error" | head -5
for g in "8 64 256 256 128 5 1 2" "4 512 128 128 128 3 1 1" "4 256 256 256 64 3 1 1" "8 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1" "4 1024 32 32 512 3 1 1"; do
  python tools/conv_bench.py $g --kinds fwd,dgrad_t 2>&1 | grep -v "^$"
done
for r in 1 2 3; do python bench.py --steps 30 --warmup 5 2>&1 | tail -1 | cut -c55-75; done
