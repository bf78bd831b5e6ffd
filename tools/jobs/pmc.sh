cd $GRAFT_REPO_ROOT
bash tools/pmc_mfma.sh r05 "bf16x3 bf16x2" > gpurun_out/pmc_log.txt 2>&1
tail -12 gpurun_out/pmc_log.txt
