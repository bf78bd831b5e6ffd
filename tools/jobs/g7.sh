cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower --showpower 2>&1 | grep -iE "power|max" | head -6 > gpurun_out/g7_sweep.txt
timeout 900 python tools/stale_skip_sweep.py >> gpurun_out/g7_sweep.txt 2>&1
cat gpurun_out/g7_sweep.txt
