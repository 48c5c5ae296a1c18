cd $GRAFT_REPO_ROOT
V=$PWD/voicebridge_amd/lib/variants
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_align.py -x -q -s > gpurun_out/r3_e2e.log 2>&1; echo "e2e+align rc=$?"; grep -E "words equal|stage errors|GPU search behind|   [A-Za-z].* [0-9]$|passed|failed|Error|^E " gpurun_out/r3_e2e.log | head -40
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_e2e.py --deselect tests/test_gpu_align.py > gpurun_out/r3_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r3_gputests.log
bash tools/ab_bench.sh product noexc product noexc
