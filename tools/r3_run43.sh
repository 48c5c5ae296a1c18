cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_config3.py tests/test_gpu_variant.py tests/test_gpu_e2e.py -m gpu -x -q > gpurun_out/r3_43_tests.log 2>&1; tail -4 gpurun_out/r3_43_tests.log
bash tools/ab_bench.sh product nosweepbits product nosweepbits
