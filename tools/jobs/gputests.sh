cd $GRAFT_REPO_ROOT
GHM_SPLIT_V1=1 GHM_SPLIT_WGRAD_V1=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "matches_oracle_at_batch_2" 2>&1 | grep -E "AssertionError: \(|passed|failed" > gpurun_out/gputests_v1.txt
GHM_SPLIT_V1=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "matches_oracle_at_batch_2" 2>&1 | grep -E "AssertionError: \(|passed|failed" >> gpurun_out/gputests_v1.txt
GHM_SPLIT_WGRAD_V1=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "matches_oracle_at_batch_2" 2>&1 | grep -E "AssertionError: \(|passed|failed" >> gpurun_out/gputests_v1.txt
cat gpurun_out/gputests_v1.txt
timeout 2400 python -m pytest tests -q -m gpu --timeout 600 2>&1 | grep -E "^FAILED|^ERROR|passed|failed|AssertionError: \(" > gpurun_out/gputests.txt
cat gpurun_out/gputests.txt
