cd $GRAFT_REPO_ROOT
bash tools/instep_repeat_sweep.sh > gpurun_out/r05_instep_repeat_sweep_bf16x3.txt 2>&1
cat gpurun_out/r05_instep_repeat_sweep_bf16x3.txt
