cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_faster.py tests/test_gpu_gmm.py -m gpu -x -q 2>&1 | tail -8
