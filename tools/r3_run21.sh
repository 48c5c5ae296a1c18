cd $GRAFT_REPO_ROOT
for m in 0 16 32 48; do VBH_GMM_PRIO=$m timeout -k 10 150 python tools/gmm_time.py > gpurun_out/gmm_prio_$m.json 2> gpurun_out/gmm_prio_$m.err || { tail -20 gpurun_out/gmm_prio_$m.err; exit 1; }; cat gpurun_out/gmm_prio_$m.json; done
