for g in "8 64 256 256 128 5 1 2" "4 512 128 128 128 3 1 1" "4 256 256 256 64 3 1 1"; do
  for ab in 0 1 2; do echo -n "ablate=$ab: "; GHM_ABLATE=$ab python tools/conv_bench.py $g --kinds fwd 2>&1 | grep -v "^$"; done
done
