cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gmm.py tests/test_gpu_e2e.py tests/test_gpu_fmllr.py -m gpu -x -q > gpurun_out/r3_22_tests.log 2>&1 ; tail -5 gpurun_out/r3_22_tests.log
for m in 0 16 32; do VBH_GMM_PRIO=$m timeout -k 10 150 python tools/gmm_time.py > gpurun_out/gmm_prio_$m.json 2> gpurun_out/gmm_prio_$m.err || { tail -20 gpurun_out/gmm_prio_$m.err; exit 1; }; cat gpurun_out/gmm_prio_$m.json; done
G=25000 timeout -k 10 150 python tools/gmm_time.py 2>/dev/null
