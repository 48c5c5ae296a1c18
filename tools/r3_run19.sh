cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_gmm.py tests/test_gpu_e2e.py -m gpu -x -q > gpurun_out/r3_19_tests.log 2>&1 ; tail -3 gpurun_out/r3_19_tests.log
python bench.py --steps 4 --warmup 1 --cpu-sample-seconds 0 > gpurun_out/r3_19_bench.json 2> gpurun_out/r3_19_bench.err && python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_19_bench.json').read().strip().splitlines()[-1])
print(d['value'], d.get('roofline_gmm'), d.get('roofline'))
PY
