cd $GRAFT_REPO_ROOT
timeout 900 python tools/mode_agreement.py 2 9 10 11 12 13 14 > gpurun_out/g5_modes.txt 2>&1
timeout 600 python tools/mode_agreement.py 4 9 10 11 >> gpurun_out/g5_modes.txt 2>&1
cat gpurun_out/g5_modes.txt
