cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_decode.py tests/test_gpu_config3.py -x -q -m gpu > gpurun_out/r2_t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t.log
bash tools/r2_probe.sh 256 > /dev/null; echo product; grep "^all utt\|^search grid" gpurun_out/r2_probe.log | cut -c1-330
python tools/variants.py run --batch 256 "$@" > gpurun_out/r2_ab.log 2>&1; grep "^==== \|^all utt\|^search grid" gpurun_out/r2_ab.log | cut -c1-330
