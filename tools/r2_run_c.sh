python -m pytest tests -m gpu -q -x -k "gpu_feat or binfuncs or bench_launch" > gpurun_out/r2_t7.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t7.log
run() { # variant sub pipeline
  VBH_LIB_PATH=$PWD/voicebridge_amd/lib/variants/libvoicebridge_hip_$1.so timeout -k 10 300 python bench.py --steps 6 --warmup 2 --sub-batches $2 --pipeline $3 --no-cpu-baseline --no-determinize > gpurun_out/r2_d_$1_s$2_p$3.json 2> gpurun_out/r2_d_$1_s$2_p$3.err; echo "bench $1 s$2 p$3 rc=$?"
  python - <<PY
import json
try:
    o=json.loads([l for l in open('gpurun_out/r2_d_$1_s$2_p$3.json') if l.startswith('{')][-1])
    print('$1 s$2 p$3', 'value %.0f'%o['value'], 'ms/step %.0f'%o['ms_per_step'], {k:(round(v,1) if not isinstance(v,(list,str)) else v) for k,v in o['search_grid_last_batch'].items() if k!='what'}, o.get('search_retries_last_batch'), {k:round(v) for k,v in o['stage_ms'].items() if k!='note'}, 'mfcc %.1f'%o['roofline_mfcc']['kernel_ms'])
except Exception as e: print('$1 s$2 p$3 failed', e); print(open('gpurun_out/r2_d_$1_s$2_p$3.err').read()[-1500:])
PY
}
run n512 1 3
run n512 2 5
run n512 4 8
run n1024 2 4
run n1024 4 8
