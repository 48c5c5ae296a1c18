cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --feat lda --steps 6 --warmup 1 --no-cpu-baseline --no-step-api --no-determinize > gpurun_out/r2_lda.json 2> gpurun_out/r2_lda.err; echo "lda rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r2_lda.json') if x.startswith('{')][-1]; j=json.loads(l)
print('lda value %.0f'%j['value'], j['config']['features'], j['config']['dim'], 'wer %.2f'%j['wer_vs_transcripts']['wer_percent'], j['search']['tokens_expanded_per_frame'])
PY
VBH_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --batch 512 --steps 3 --warmup 1 --no-cpu-baseline --no-step-api --no-determinize > gpurun_out/r2_g2.json 2> gpurun_out/r2_g2.err; echo "gpus2 rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r2_g2.json') if x.startswith('{')][-1]; j=json.loads(l)
print('gpus2 (two ranks on one device, gloo) value %.0f'%j['value'], 'n_gpus', j['n_gpus'], j.get('gather',{}).get('counters'), 'wer %.2f'%j['wer_vs_transcripts']['wer_percent'])
PY
