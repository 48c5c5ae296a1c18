cd $GRAFT_REPO_ROOT
VBH_PROBE_REPEAT=5 timeout -k 10 500 python bench.py --probe --batch 1024 --no-cpu-baseline --no-step-api > gpurun_out/r3_40_probe.log 2>&1; grep -a "determinism\|differ\|repeat" gpurun_out/r3_40_probe.log | head -5
VBH_PROBE_REPEAT=4 VBH_LIB_PATH=$GRAFT_REPO_ROOT/voicebridge_amd/lib/variants/libvoicebridge_hip_nt512.so timeout -k 10 500 python bench.py --probe --batch 1024 --no-cpu-baseline --no-step-api > gpurun_out/r3_40_probe512.log 2>&1; grep -a "determinism\|differ\|repeat" gpurun_out/r3_40_probe512.log | head -5
bash tools/soak.sh 2
