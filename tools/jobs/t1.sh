cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_fullsize.py -q -m gpu --timeout 600 2>&1 | tail -4
