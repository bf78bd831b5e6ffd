for g in "4 512 128 128 128 3 1 1" "4 256 256 256 128 3 1 1" "4 1024 64 64 256 3 1 1" "4 128 256 256 128 3 1 1"; do
  python tools/conv_bench.py $g --kinds wgrad 2>&1 | grep -v "^$"
  GHM_WGRAD_BM256=1 python tools/conv_bench.py $g --kinds wgrad 2>&1 | grep -v "^$"
done
GHM_WGRAD_BM256=1 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_fwd_dgrad or fullsize" 2>&1 | tail -2
GHM_WGRAD_BM256=1 python -m pytest tests/test_gpu_fullsize.py -x -q -k "sampled" 2>&1 | tail -2
