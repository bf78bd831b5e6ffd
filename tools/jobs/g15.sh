cd $GRAFT_REPO_ROOT
for g in "4 128 256 256 64 3 1 1" "4 1024 64 64 64 3 1 1"; do
for abl in 0 1 2 7; do
  echo -n "$g ablate $abl: "
  GHM_SPLIT_ABLATE=$abl timeout 120 python tools/conv_bench.py $g --kinds fwd --reps 30 --dtype split --q q | awk '{printf "%s %s %s | ", $1, $3, $5}'; echo
done; done > gpurun_out/g15.txt 2>&1
cat gpurun_out/g15.txt
