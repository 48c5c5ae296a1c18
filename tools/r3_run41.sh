cd $GRAFT_REPO_ROOT
export VBH_LIB_PATH=$GRAFT_REPO_ROOT/voicebridge_amd/lib/variants/libvoicebridge_hip_fused.so
timeout -k 10 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_config3.py -m gpu -x -q > gpurun_out/r3_41_tests.log 2>&1; tail -5 gpurun_out/r3_41_tests.log
unset VBH_LIB_PATH
bash tools/ab_bench.sh product fused product fused
