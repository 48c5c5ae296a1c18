# Same-box sweep of the pruning interval (VBH_PRUNE_EVERY; under gpurun): bash tools/prune_every.sh 100 150 180
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  VBH_PRUNE_EVERY=$v timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-step-api --no-determinize > gpurun_out/pe_$v.json 2> gpurun_out/pe_$v.err; echo "every $v rc=$?"
  python - <<PY
import json
l=[x for x in open('gpurun_out/pe_$v.json') if x.startswith('{')][-1]; j=json.loads(l)
print('prune every $v', 'value %.0f'%j['value'], 'kernel_ms %.0f'%j['roofline']['kernel_ms'], j['search_grid_last_batch']['utterance_ms_min_mean_max'], j['search_retries_last_batch'])
PY
done
