set -o pipefail
python -m pytest tests -m gpu -q -x -k "steps or fmllr or binfuncs or decodable" > gpurun_out/r2_t11.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r2_t11.log
VBH_DEBUG_TIMES=1 timeout -k 10 400 python tools/bench_steps.py --utts 1024 --nj 4 --fmllr > gpurun_out/r2_steps.json 2> gpurun_out/r2_steps.err; echo "steps rc=$?"; grep -a "^DecodeFmllr" gpurun_out/r2_steps.err | head -20; tail -c 900 gpurun_out/r2_steps.json
