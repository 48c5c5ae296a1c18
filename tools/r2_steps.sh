cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/bench_steps.py --utts ${1:-256} --nj 4 --fmllr > gpurun_out/r2_steps.json 2> gpurun_out/r2_steps.err; echo "steps rc=$?"
tail -15 gpurun_out/r2_steps.err; cat gpurun_out/r2_steps.json
