# tuning sweep of the weight-gradient patch kernels (isolated rates; GHM_ABLATE = staging ablation: 1 no global loads, 2 no LDS stores either)
python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -3
for ab in 0 1; do
  echo "== GHM_ABLATE=$ab"
  for g in "8 64 256 256 128 3 2 1" "8 128 128 128 256 3 2 1" "4 256 64 64 512 3 2 1" "4 512 128 128 128 3 1 1" "4 1024 64 64 256 3 1 1" "4 256 256 256 64 3 1 1" "8 64 256 256 128 5 1 2" "8 128 128 128 128 5 1 2"; do
    GHM_ABLATE=$ab python tools/conv_bench.py $g --kinds wgrad --reps 20
  done
done
