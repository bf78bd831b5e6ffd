set -x
python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -5
for g in "4 64 256 256 128 3 2 1" "8 64 256 256 128 3 2 1" "4 128 128 128 256 3 2 1" "8 128 128 128 256 3 2 1" "4 256 64 64 512 3 2 1" "8 256 64 64 512 3 2 1"; do
  python tools/conv_bench.py $g --kinds dgrad_t,wgrad,fwd --reps 20
done
