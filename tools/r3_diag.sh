cd $GRAFT_REPO_ROOT
V=$PWD/voicebridge_amd/lib/variants
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py -x -q -s > gpurun_out/r3_e2e.log 2>&1; echo "e2e rc=$?"; grep -E "words equal|stage errors|GPU search behind|   [A-Za-z].* [0-9]$|passed|failed|Error|assert" gpurun_out/r3_e2e.log | head -40
for v in product diag diag0h diag1h diag2h diag3h; do
  if [ "$v" = product ]; then unset VBH_LIB_PATH; else export VBH_LIB_PATH=$V/libvoicebridge_hip_$v.so; fi
  echo "==== $v"; timeout -k 10 300 python bench.py --probe --batch 256 --no-cpu-baseline 2>&1 | grep -E "decode of|search grid|all utterances|phase share|VBH_DIAG|frames by tokens" | head -14
done
