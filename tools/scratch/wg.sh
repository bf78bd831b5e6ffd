for g in "4 512 128 128 128 3 1 1" "8 64 256 256 128 5 1 2" "4 256 256 256 64 3 1 1" "4 128 256 256 256 3 2 1" "8 128 128 128 128 5 1 2"; do
  python tools/conv_bench.py $g --kinds wgrad 2>&1 | grep -v "^$"
done
