set -o pipefail
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t10.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2_t10.log
timeout -k 10 500 python bench.py --steps 8 --warmup 2 > gpurun_out/r2_n.json 2> gpurun_out/r2_n.err; rc=$?; echo "bench rc=$rc"; grep -c "Memory access fault" gpurun_out/r2_n.err
if [ $rc -eq 0 ]; then python - <<PY
import json
o=json.loads([l for l in open('gpurun_out/r2_n.json') if l.startswith('{')][-1])
for k in ['value','ms_per_step','stage_ms','roofline','roofline_gmm','roofline_mfcc','cpu_baseline','cpu_baseline_1proc','parity_words_equal','search','search_grid_last_batch','search_retries_last_batch','lattice_stage_in_step']:
    print(k, json.dumps(o.get(k))[:600])
PY
fi
