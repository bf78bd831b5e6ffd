cd $GRAFT_REPO_ROOT
for g in "4 256 128 128 128 3 1 1" "4 128 256 256 64 3 1 1" "4 1024 64 64 256 3 1 1" "4 512 128 128 128 3 1 1"; do
  echo "== $g"
  for env in "X=1" "GHM_SPLIT_W8=1" "GHM_SPLIT_BM64=1" "GHM_SPLIT_V1=1"; do
    echo -n "$env: "; env $env timeout 120 python tools/conv_bench.py $g --kinds fwd,dgrad_t --reps 30 --dtype split --q q | awk '{printf "%s %s %s | ", $1, $3, $5}'; echo
  done
done > gpurun_out/g3.txt 2>&1
cat gpurun_out/g3.txt
