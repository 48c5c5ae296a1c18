cd $GRAFT_REPO_ROOT
V=$PWD/voicebridge_amd/lib/variants
probe() { echo "==== $1 $2"; env $2 VBH_LIB_PATH=$V/libvoicebridge_hip_$1.so VBH_PROBE_REPEAT=6 timeout -k 10 400 python bench.py --probe --batch 1024 --no-cpu-baseline 2>&1 | grep -E "determinism|search grid|repeat" | cut -c1-200; }
probe nt512 X=1
timeout -k 10 600 python -m pytest tests/test_gpu_config3.py tests/test_gpu_decode.py tests/test_gpu_e2e.py -q -x -m gpu 2>&1 | tail -3
VBH_LIB_PATH=$V/libvoicebridge_hip_nt512.so timeout -k 10 600 python -m pytest tests/test_gpu_config3.py tests/test_gpu_decode.py -q -x -m gpu 2>&1 | tail -3
bash tools/ab_bench.sh product
