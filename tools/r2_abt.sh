cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_config3.py -x -q -m gpu > gpurun_out/r2_t.log 2>&1; echo "product pytest rc=$?"; tail -2 gpurun_out/r2_t.log
bash tools/r2_abp.sh "$@"
