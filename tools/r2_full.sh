cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python bench.py "$@" > gpurun_out/r2_full.json 2> gpurun_out/r2_full.err; echo "bench rc=$?"
tail -5 gpurun_out/r2_full.err
