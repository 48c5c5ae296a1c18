cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2_t.log 2>&1; echo "product pytest rc=$?"; tail -2 gpurun_out/r2_t.log
bash tools/ab_bench.sh "$@"
