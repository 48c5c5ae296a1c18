cd $GRAFT_REPO_ROOT
export VBH_LIB_PATH=$GRAFT_REPO_ROOT/voicebridge_amd/lib/variants/libvoicebridge_hip_nt512.so
timeout -k 10 1000 python -m pytest tests/test_gpu_decode.py tests/test_gpu_config3.py tests/test_gpu_e2e.py tests/test_gpu_faster.py tests/test_gpu_align.py -m gpu -x -q > gpurun_out/r3_45_tests.log 2>&1; tail -4 gpurun_out/r3_45_tests.log
VBH_PROBE_REPEAT=4 timeout -k 10 500 python bench.py --probe --batch 1024 --no-cpu-baseline --no-step-api > gpurun_out/r3_45_probe512.log 2>&1; grep -a "determinism" gpurun_out/r3_45_probe512.log
