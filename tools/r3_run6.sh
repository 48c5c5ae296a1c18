cd $GRAFT_REPO_ROOT
timeout -k 10 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -8 gpurun_out/r3_gputests.log
VBH_DEBUG_TIMES=1 timeout -k 10 600 python tools/bench_steps.py --utts 1024 --nj 4 --fmllr > gpurun_out/r3_steps.json 2> gpurun_out/r3_steps.err; echo "steps rc=$?"; cat gpurun_out/r3_steps.json | cut -c1-1200
timeout -k 10 900 python bench.py --steps 8 --warmup 2 > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r3_bench.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3_bench.json') if l.startswith('{')][-1])
for k in ('value','ms_per_step','vs_baseline'): print(k, j.get(k))
print('roofline', {k:j['roofline'][k] for k in ('frac','kernel_ms','achieved')})
print('gmm', {k:j['roofline_gmm'].get(k) for k in ('frac','kernel_ms','achieved','traffic')})
print('raw', j.get('raw_lattice_rate',{}).get('value'), 'lat', j.get('lattice_stage_in_step',{}).get('worker_ms_per_utterance'))
print('cpu', j.get('cpu_baseline'), j.get('parity_words_equal'), j.get('parity_wer_on_the_cpu_sample'))
print('step_api', j.get('step_api'))
print('ranks', j.get('ranks'))
print('wer', j.get('wer_vs_transcripts',{}).get('wer_percent'))
PY
