python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad" 2>&1 | tail -2
for g in "8 64 256 256 128 5 1 2" "8 128 128 128 128 5 1 2" "4 512 128 128 128 3 1 1" "4 256 256 256 64 3 1 1" "8 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1" "8 256 64 64 512 3 2 1"; do
  python tools/conv_bench.py $g --kinds wgrad 2>&1 | grep -v "^$"
  case "$g" in *"3 2 1") echo -n "bkp16: "; GHM_WGRAD_BKP16=1 python tools/conv_bench.py $g --kinds wgrad 2>&1 | grep -v "^$";; esac
done
