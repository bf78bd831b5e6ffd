cd $GRAFT_REPO_ROOT
g="8 128 128 128 128 5 1 2"
for abl in 0 1 8 16; do
  echo "== ablate $abl"
  GHM_SPLIT_ABLATE=$abl timeout 120 python tools/conv_bench.py $g --kinds fwd --reps 30 --dtype split --q q | tr '\n' '|'; echo
  GHM_BENCH_ZERO=1 GHM_SPLIT_ABLATE=$abl timeout 120 python tools/conv_bench.py $g --kinds fwd --reps 30 --dtype split --q q | tr '\n' '|'; echo
done > gpurun_out/g2_ablate2.txt 2>&1
cat gpurun_out/g2_ablate2.txt
