cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r03_e_gpu_tests.log 2>&1; tail -3 gpurun_out/r03_e_gpu_tests.log
bash tools/profile_bench.sh r3e
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r03_e_bench_driver_style.json 2> gpurun_out/r03_e_bench_driver_style.err; tail -c 600 gpurun_out/r03_e_bench_driver_style.json
