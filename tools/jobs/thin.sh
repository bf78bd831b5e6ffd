cd $GRAFT_REPO_ROOT
run() { echo -n "$1 | $2: "; env $1 timeout 120 python tools/conv_bench.py $2 --kinds $3 --reps 50 2>&1 | tail -2 | tr '\n' ' '; echo; }
G1="8 4 512 512 64 3 2 1"
G2="4 1 512 512 64 3 2 1"
for e in "X=0" "GHM_FANOUT_RPI=1" "GHM_FANOUT_RPI=4" "GHM_FANOUT_RPI=8" "GHM_FANOUT_BPC=2" "GHM_FANOUT_BPC=2 GHM_FANOUT_RPI=4" "GHM_FANOUT_SPLIT_ALL=1" "GHM_NO_THIN=1"; do
  run "$e" "$G1" fwd
done
for e in "X=0" "GHM_FANOUT_RPI=4" "GHM_FANOUT_BPC=2"; do run "$e" "$G2" fwd; done
run "X=0" "$G1" wgrad
run "X=0" "8 1 512 512 64 5 1 2" fwd
run "X=0" "4 64 256 256 4 3 1 1" fwd,dgrad,wgrad
