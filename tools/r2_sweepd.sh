cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py --batch 256 --steps 1 --warmup 0 --no-cpu-baseline --no-step-api --no-determinize --sweep-depth ${1:-0.08,0.12,0.16,0.2} > gpurun_out/r2_sweepd.json 2> gpurun_out/r2_sweepd.err; echo "sweep rc=$?"
tail -3 gpurun_out/r2_sweepd.err
