cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_config3.py -x -q -m gpu > gpurun_out/r2_t.log 2>&1; echo "product pytest rc=$?"; tail -2 gpurun_out/r2_t.log
for v in n768 n512; do VBH_LIB_PATH=$GRAFT_REPO_ROOT/voicebridge_amd/lib/variants/libvoicebridge_hip_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_config3.py -x -q -m gpu > gpurun_out/r2_t_$v.log 2>&1; echo "$v pytest rc=$?"; tail -2 gpurun_out/r2_t_$v.log; done
bash tools/r2_abq.sh old product n768 n512
