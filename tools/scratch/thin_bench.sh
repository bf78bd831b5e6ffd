for ab in 0 8; do echo -n "ablate $ab: "; GHM_ABLATE=$ab python tools/conv_bench.py 8 1 512 512 64 5 1 2 --kinds fwd 2>&1 | grep -v "^$"; done
for b in 128 256 512; do echo -n "blocks $b: "; GHM_THIN_BLOCKS=$b python tools/conv_bench.py 8 1 512 512 64 5 1 2 --kinds fwd 2>&1 | grep -v "^$"; done
