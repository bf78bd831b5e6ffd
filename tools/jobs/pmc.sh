cd $GRAFT_REPO_ROOT
bash tools/pmc_mfma.sh r05 "bf16x3" > gpurun_out/pmc_log.txt 2>&1
tail -6 gpurun_out/pmc_log.txt
