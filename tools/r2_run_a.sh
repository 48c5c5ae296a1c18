set -x
python -m pytest tests -m gpu -q -x -k "gpu_decode or gpu_config3 or gpu_feat" > gpurun_out/r2_t5.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t5.log
VBH_LIB_PATH=$PWD/voicebridge_amd/lib/variants/libvoicebridge_hip_n512.so python -m pytest tests -m gpu -q -x -k "gpu_decode or gpu_config3" > gpurun_out/r2_t5b.log 2>&1; echo "pytest n512 rc=$?"; tail -3 gpurun_out/r2_t5b.log
for v in n1024 n512; do for p in 2 3; do
  VBH_LIB_PATH=$PWD/voicebridge_amd/lib/variants/libvoicebridge_hip_$v.so timeout -k 10 400 python bench.py --steps 6 --warmup 2 --pipeline $p --no-cpu-baseline --no-determinize > gpurun_out/r2_b_${v}_p$p.json 2> gpurun_out/r2_b_${v}_p$p.err; echo "bench $v p$p rc=$?"
  python - <<PY
import json
try:
    o=json.loads([l for l in open('gpurun_out/r2_b_${v}_p$p.json') if l.startswith('{')][-1])
    print('$v p$p', 'value %.0f'%o['value'], 'ms/step %.0f'%o['ms_per_step'], o['search_grid_last_batch'], {k:round(v) for k,v in o['stage_ms'].items() if k!='note'}, o['roofline_mfcc']['kernel_ms'])
except Exception as e: print('$v p$p failed', e)
PY
done; done
