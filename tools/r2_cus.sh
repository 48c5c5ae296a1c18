cd $GRAFT_REPO_ROOT
for c in 224 232 240 248; do
  timeout -k 10 400 python bench.py --steps 6 --warmup 1 --search-cus $c --no-cpu-baseline --no-step-api --no-determinize > gpurun_out/r2_cus_$c.json 2> gpurun_out/r2_cus_$c.err; echo "cus $c rc=$?"
  python - <<PY
import json
l=[x for x in open('gpurun_out/r2_cus_$c.json') if x.startswith('{')][-1]; j=json.loads(l)
print('search-cus $c', 'value %.0f'%j['value'], 'kernel_ms %.0f'%j['roofline']['kernel_ms'], 'gmm in pipeline %.0f'%j['stage_ms']['gmm'], j['search_grid_last_batch']['utterance_ms_min_mean_max'])
PY
done
