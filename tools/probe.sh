# bench.py --probe on a batch (default 256): one search, its grid statistics and phase shares (under gpurun)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --batch ${1:-256} --steps 1 --warmup 0 --no-cpu-baseline --no-step-api --no-determinize --probe > gpurun_out/r2_probe.log 2>&1; echo "probe rc=$?"
tail -14 gpurun_out/r2_probe.log
