run() { # variant sub pipeline ahead cus
  VBH_LIB_PATH=$PWD/voicebridge_amd/lib/variants/libvoicebridge_hip_$1.so timeout -k 10 300 python bench.py --steps 6 --warmup 2 --sub-batches $2 --pipeline $3 --ahead $4 --search-cus $5 --no-cpu-baseline --no-determinize > gpurun_out/r2_e.json 2> gpurun_out/r2_e.err; echo "bench $* rc=$?"
  python - <<PY
import json
try:
    o=json.loads([l for l in open('gpurun_out/r2_e.json') if l.startswith('{')][-1])
    print('$*', 'value %.0f'%o['value'], 'ms/step %.0f'%o['ms_per_step'], o['search_grid_last_batch']['utterance_ms_min_mean_max'], {k:round(v) for k,v in o['stage_ms'].items() if k!='note'})
except Exception as e: print('$* failed', e); print(open('gpurun_out/r2_e.err').read()[-1500:])
PY
}
run n512 1 3 1 208
run n512 1 3 1 192
run n512 1 3 2 208
run n512 2 4 2 192
run n512 2 6 3 192
run n1024 1 2 1 208
