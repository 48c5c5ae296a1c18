cd $GRAFT_REPO_ROOT
export VBH_LIB_PATH=$GRAFT_REPO_ROOT/voicebridge_amd/lib/variants/libvoicebridge_hip_diag.so
timeout -k 10 300 python bench.py --probe --batch 256 --no-cpu-baseline --no-step-api 2>&1 | grep -a -E "decode of|search grid|all utterances|phase share|VBH_DIAG|frames by tokens|phase" | head -20 | cut -c1-1200
