cd $GRAFT_REPO_ROOT
LD=2528 timeout -k 10 150 python tools/gmm_time.py 2>gpurun_out/gmm_time.err || tail -5 gpurun_out/gmm_time.err
python -m pytest tests/test_gpu_gmm.py -m gpu -x -q 2>&1 | tail -2
bash tools/r3_run29.sh | tail -4
