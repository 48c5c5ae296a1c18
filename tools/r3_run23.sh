cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r3_23_tests.log 2>&1 ; tail -5 gpurun_out/r3_23_tests.log
python bench.py --steps 4 --warmup 1 --cpu-sample-seconds 0 > gpurun_out/r3_23_bench.json 2> gpurun_out/r3_23_bench.err && python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_23_bench.json').read().strip().splitlines()[-1])
print(d['value'], {k:d['roofline_gmm'][k] for k in ('achieved','frac','kernel_ms','in_pipeline_ms')}, {k:d['roofline'][k] for k in ('achieved','frac','kernel_ms')})
PY
