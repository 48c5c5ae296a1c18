python -m pytest tests -m gpu -q -x -k "gpu_gmm or gpu_feat or gpu_decode or config3" > gpurun_out/r2_t9.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t9.log
timeout -k 10 240 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-determinize > gpurun_out/r2_g.json 2> gpurun_out/r2_g.err; rc=$?; echo "bench rc=$rc"; tail -c 600 gpurun_out/r2_g.err
if [ $rc -eq 0 ]; then python - <<PY
import json
o=json.loads([l for l in open('gpurun_out/r2_g.json') if l.startswith('{')][-1])
for k in ['value','ms_per_step','stage_ms','roofline_gmm','roofline_mfcc','search','search_grid_last_batch','search_retries_last_batch']:
    print(k, json.dumps(o.get(k))[:600])
PY
fi
