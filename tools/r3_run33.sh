cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_faster.py tests/test_gpu_align.py -m gpu -x -q 2>&1 | tail -3
VBH_PROBE_FASTER=1 timeout -k 10 400 python bench.py --probe --batch 1024 --no-cpu-baseline --no-step-api > gpurun_out/r3_33_f.log 2>&1; grep -a "FasterDecoder" gpurun_out/r3_33_f.log
