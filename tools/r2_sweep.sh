cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_binfuncs.py tests/test_gpu_steps.py -x -q -m gpu > gpurun_out/r2_t14.log 2>&1; echo "pytest rc=$?"
timeout -k 10 600 python bench.py --batch 256 --steps 1 --warmup 0 --no-cpu-baseline --no-step-api --no-determinize --sweep-mean-scale 1.0,1.15,1.3,1.5,2.0 > gpurun_out/r2_sweep228.json 2> gpurun_out/r2_sweep228.err; echo "sweep rc=$?"
