cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|AssertionError|Error" | head -40 > gpurun_out/gputests.txt
cat gpurun_out/gputests.txt
