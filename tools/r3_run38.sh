cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_steps.py tests/test_gpu_binfuncs.py tests/test_gpu_fmllr.py tests/test_gpu_align.py -m gpu -x -q 2>&1 | tail -3
VBH_DEBUG_TIMES=1 timeout -k 10 600 python tools/bench_steps.py --utts 1024 --nj 4 --fmllr > gpurun_out/r3_38_steps.json 2> gpurun_out/r3_38_steps.err; tail -2 gpurun_out/r3_38_steps.json
grep -a "lap read graph\|lap device graph\|lap features\|lap decoder" gpurun_out/r3_38_steps.err | head -12
timeout -k 10 600 python tools/bench_steps.py --utts 1024 --nj 4 --fmllr 2>/dev/null | tail -1
