python -m pytest tests/test_gpu_ops.py -x -q -k "thin" 2>&1 | tail -2
for g in "8 1 512 512 64 5 1 2" "8 4 512 512 64 3 2 1" "4 1 512 512 64 3 2 1" "4 3 512 512 128 2 2 0" "4 64 512 512 1 5 1 2" "4 64 256 256 4 3 1 1"; do
  python tools/conv_bench.py $g --kinds wgrad 2>&1 | grep -v "^$"
done
