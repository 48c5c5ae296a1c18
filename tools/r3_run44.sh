cd $GRAFT_REPO_ROOT
for pe in 300 200 140 450; do
  VBH_PRUNE_EVERY=$pe timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-step-api --no-determinize > gpurun_out/r3_44.json 2> gpurun_out/r3_44.err; echo "prune_every=$pe rc=$?"
  python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3_44.json') if x.startswith('{')][-1]; j=json.loads(l)
print('value %.0f'%j['value'], 'kernel_ms %.0f'%j['roofline']['kernel_ms'], j['search_grid_last_batch']['utterance_ms_min_mean_max'], j['search_retries_last_batch'])
PY
done
