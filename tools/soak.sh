# Repeat the default bench (under gpurun) and report value / retries / rc of every run: bash tools/soak.sh 4
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-3}); do
  timeout -k 10 400 python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-step-api > gpurun_out/soak_$i.json 2> gpurun_out/soak_$i.err; rc=$?
  python - <<PY
import json
try:
    l=[x for x in open('gpurun_out/soak_$i.json') if x.startswith('{')][-1]; j=json.loads(l)
    print('run $i rc=$rc value %.0f'%j['value'], j['search_retries_last_batch'], 'wer %.2f'%j['wer_vs_transcripts']['wer_percent'])
except Exception as e:
    print('run $i rc=$rc NO JSON', e)
PY
  grep -i "fault\|error\|WARNING" gpurun_out/soak_$i.err | head -3
done
