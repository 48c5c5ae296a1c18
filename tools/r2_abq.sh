cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = product ]; then unset VBH_LIB_PATH; else export VBH_LIB_PATH=$GRAFT_REPO_ROOT/voicebridge_amd/lib/variants/libvoicebridge_hip_$v.so; fi
  timeout -k 10 200 python bench.py --batch 256 --steps 1 --warmup 0 --no-cpu-baseline --no-step-api --no-determinize --probe > gpurun_out/r2_abq_$v.log 2>&1; echo "$v probe256: $(grep '^search grid' gpurun_out/r2_abq_$v.log | cut -c1-160)"
  timeout -k 10 200 python bench.py --batch 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-step-api --no-determinize --probe > gpurun_out/r2_abq_$v.log 2>&1; echo "$v probe1024: $(grep '^search grid' gpurun_out/r2_abq_$v.log | cut -c1-160)"
  timeout -k 10 400 python bench.py --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline --no-step-api --no-determinize > gpurun_out/r2_abq_$v.json 2> gpurun_out/r2_abq_$v.err
  python - <<PY
import json
l=[x for x in open('gpurun_out/r2_abq_$v.json') if x.startswith('{')][-1]; j=json.loads(l)
print('$v pipeline1', 'value %.0f'%j['value'], 'kernel_ms %.0f'%j['roofline']['kernel_ms'], j['search_grid_last_batch']['utterance_ms_min_mean_max'], j['search_grid_last_batch'].get('workgroups'))
PY
done
