# Same-box A/B of library variants built by tools/variants.py (under gpurun): bash tools/ab_bench.sh product <name> <name> ...
# Each runs the default bench for 4 steps; prints value, search kernel time and per-utterance search time.
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = product ]; then unset VBH_LIB_PATH; else export VBH_LIB_PATH=$GRAFT_REPO_ROOT/voicebridge_amd/lib/variants/libvoicebridge_hip_$v.so; fi
  timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-step-api --no-determinize > gpurun_out/r2_abp_$v.json 2> gpurun_out/r2_abp_$v.err; echo "$v rc=$?"
  python - <<PY
import json
l=[x for x in open('gpurun_out/r2_abp_$v.json') if x.startswith('{')][-1]; j=json.loads(l)
print('$v', 'value %.0f'%j['value'], 'kernel_ms %.0f'%j['roofline']['kernel_ms'], j['search_grid_last_batch']['utterance_ms_min_mean_max'], j['search_retries_last_batch'])
PY
done
