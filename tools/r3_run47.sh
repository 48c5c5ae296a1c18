cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_steps.py tests/test_gpu_fmllr.py -m gpu -x -q 2>&1 | tail -2
VBH_DEBUG_TIMES=1 timeout -k 10 600 python tools/bench_steps.py --utts 1024 --nj 4 --fmllr 2> gpurun_out/r3_47.err | tail -1 | cut -c1-900
grep -a "^DecodeFmllr weight-silence\|^DecodeFmllr gmm-est\|^DecodeFmllr gmm-rescore" gpurun_out/r3_47.err | sort | awk '{a[$2]=a[$2]" "$3} END{for(k in a) print k, a[k]}'
