cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_decode.py -m gpu -x -q -k "small_shared_pool" 2>&1 | tail -5
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_37_tests.log 2>&1; tail -3 gpurun_out/r3_37_tests.log
