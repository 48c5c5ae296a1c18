cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_align.py -x -q -s > gpurun_out/r3_e2e.log 2>&1; echo "e2e+align rc=$?"; grep -E "words equal|stage errors|GPU search behind|   [A-Za-z].* [0-9]$|passed|failed|Error|^E " gpurun_out/r3_e2e.log | head -40
VBH_DEBUG_TIMES=1 timeout -k 10 600 python tools/bench_steps.py --utts 1024 --nj 4 --fmllr > gpurun_out/r3_steps.json 2> gpurun_out/r3_steps.err; echo "steps rc=$?"; cat gpurun_out/r3_steps.json | cut -c1-1500
VBH_PROBE_REPEAT=4 timeout -k 10 600 python bench.py --probe --batch 1024 --no-cpu-baseline 2>&1 | grep -E "repeat|determinism|search grid" | head
