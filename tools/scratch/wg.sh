for g in "4 512 128 128 128 3 1 1" "8 64 256 256 128 5 1 2"; do
  python tools/conv_bench.py $g --kinds wgrad 2>&1 | grep -v "^$"
  for l in 41 14; do echo -n "layout $l: "; GHM_WG_LAYOUT=$l python tools/conv_bench.py $g --kinds wgrad 2>&1 | grep -v "^$"; done
done
