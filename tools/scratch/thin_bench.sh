python -m pytest tests/test_gpu_ops.py -x -q -k "thin or taps" > gpurun_out/t.txt 2>&1; tail -5 gpurun_out/t.txt
python tools/conv_bench.py 4 64 512 512 1 5 1 2 --kinds fwd 2>&1 | grep -v "^$"
python tools/conv_bench.py 4 1 512 512 64 5 1 2 --kinds dgrad 2>&1 | grep -v "^$"
