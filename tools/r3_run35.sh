cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_decode.py tests/test_gpu_config3.py tests/test_gpu_variant.py tests/test_gpu_e2e.py -m gpu -x -q > gpurun_out/r3_35_tests.log 2>&1; tail -3 gpurun_out/r3_35_tests.log
python bench.py --steps 4 --warmup 1 --cpu-sample-seconds 0 --no-step-api > gpurun_out/r3_35_bench.json 2> gpurun_out/r3_35_bench.err && python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_35_bench.json').read().strip().splitlines()[-1])
print(d['value'], {k:d['roofline'][k] for k in ('achieved','frac','kernel_ms')})
PY
