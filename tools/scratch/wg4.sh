python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -2
for g in "8 64 256 256 128 5 1 2" "4 512 128 128 128 3 1 1" "4 256 256 256 64 3 1 1" "8 64 256 256 128 3 2 1" "4 1024 64 64 256 3 1 1"; do
  python tools/conv_bench.py $g 2>&1 | grep -v "^$"
done
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-400
