# The GPU test suite, then the default bench (under gpurun); outputs in gpurun_out/r2_t.log and gpurun_out/r2_full.json
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2_t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t.log
timeout -k 10 1000 python bench.py > gpurun_out/r2_full.json 2> gpurun_out/r2_full.err; echo "bench rc=$?"; tail -3 gpurun_out/r2_full.err
