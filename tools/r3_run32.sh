cd $GRAFT_REPO_ROOT
VBH_PROBE_FASTER=1 VBH_LIB_PATH=$GRAFT_REPO_ROOT/voicebridge_amd/lib/variants/libvoicebridge_hip_fprof.so timeout -k 10 400 python bench.py --probe --batch 1024 --no-cpu-baseline --no-step-api > gpurun_out/r3_32_fprof.log 2>&1; grep -a "faster utt\|FasterDecoder" gpurun_out/r3_32_fprof.log | head -8
VBH_PROBE_FASTER=1 timeout -k 10 400 python bench.py --probe --batch 1024 --no-cpu-baseline --no-step-api > gpurun_out/r3_32_f.log 2>&1; grep -a "FasterDecoder" gpurun_out/r3_32_f.log
