cd $GRAFT_REPO_ROOT
for opt in "--pipeline 2" "--pipeline 3" "--pipeline 2 --ahead 2"; do
  timeout -k 10 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-step-api $opt > gpurun_out/r3_39.json 2> gpurun_out/r3_39.err; echo "$opt rc=$?"
  python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3_39.json') if x.startswith('{')][-1]; j=json.loads(l)
print('value %.0f'%j['value'], 'ms/step %.1f'%j['ms_per_step'], 'kernel_ms %.0f'%j['roofline']['kernel_ms'], 'gmm in pipeline %.1f'%j['roofline_gmm']['in_pipeline_ms'])
PY
done
