python -m pytest tests -m gpu -q -x > gpurun_out/r2_t8.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2_t8.log
timeout -k 10 600 python bench.py --steps 8 --warmup 2 > gpurun_out/r2_f.json 2> gpurun_out/r2_f.err; echo "bench rc=$?"; tail -c 400 gpurun_out/r2_f.err
python - <<PY
import json
o=json.loads([l for l in open('gpurun_out/r2_f.json') if l.startswith('{')][-1])
for k in ['value','ms_per_step','stage_ms','roofline_gmm','roofline_mfcc','cpu_baseline','cpu_baseline_1proc','parity_words_equal','search','search_grid_last_batch','lattice_stage_in_step']:
    print(k, json.dumps(o.get(k))[:700])
PY
