cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gmm.py -m gpu -x -q -rs > gpurun_out/r3_24_tests.log 2>&1 ; tail -3 gpurun_out/r3_24_tests.log
timeout -k 10 150 python tools/gmm_time.py 2>gpurun_out/gmm_time.err || tail -5 gpurun_out/gmm_time.err
G=25000 timeout -k 10 150 python tools/gmm_time.py 2>gpurun_out/gmm_time.err || tail -5 gpurun_out/gmm_time.err
