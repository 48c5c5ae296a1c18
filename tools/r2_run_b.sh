for p in 2 3; do
  VBH_LIB_PATH=$PWD/voicebridge_amd/lib/variants/libvoicebridge_hip_n512.so timeout -k 10 500 python bench.py --steps 6 --warmup 2 --pipeline $p --no-cpu-baseline --no-determinize > gpurun_out/r2_c_n512_p$p.json 2> gpurun_out/r2_c_n512_p$p.err; echo "bench n512 p$p rc=$?"
  python - <<PY
import json
try:
    o=json.loads([l for l in open('gpurun_out/r2_c_n512_p$p.json') if l.startswith('{')][-1])
    print('n512 p$p', 'value %.0f'%o['value'], 'ms/step %.0f'%o['ms_per_step'], o['search_grid_last_batch'], o.get('search_retries_last_batch'), {k:round(v) for k,v in o['stage_ms'].items() if k!='note'})
except Exception as e: print('n512 p$p failed', e)
PY
done
python -m pytest tests -m gpu -q -x -k "gpu_decode or gpu_config3 or decodable or gpu_steps" > gpurun_out/r2_t6.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t6.log
