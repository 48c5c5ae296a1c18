cd $GRAFT_REPO_ROOT
VBH_BENCH_BACKEND=gloo timeout -k 10 700 python bench.py --gpus 2 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline --no-step-api > gpurun_out/r3_g2.json 2> gpurun_out/r3_g2.err; echo "gpus2 rc=$?"
tail -c 1500 gpurun_out/r3_g2.json
tail -5 gpurun_out/r3_g2.err
