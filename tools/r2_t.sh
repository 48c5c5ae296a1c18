cd $GRAFT_REPO_ROOT
timeout -k 10 ${2:-900} python -m pytest $1 -x -q -m gpu > gpurun_out/r2_t.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r2_t.log
