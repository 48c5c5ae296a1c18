cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout -k 10 1100 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3_bench_final.json') if l.startswith('{')][-1])
for k in ('value','ms_per_step'): print(k, j.get(k))
print('roofline', {k:j['roofline'].get(k) for k in ('kernel','frac','kernel_ms','achieved','traffic')})
print('gmm', {k:j['roofline_gmm'].get(k) for k in ('frac','kernel_ms','achieved','traffic')})
print('mfcc', {k:j['roofline_mfcc'].get(k) for k in ('frac','kernel_ms','traffic')})
print('raw', j.get('raw_lattice_rate',{}).get('value'))
print('cpu', j.get('cpu_baseline',{}).get('value'), j.get('cpu_baseline',{}).get('cores'), j.get('cpu_baseline_1proc',{}).get('value'), j.get('parity_words_equal'), j.get('parity_wer_on_the_cpu_sample'))
print('step_api', {k:v for k,v in j.get('step_api',{}).items() if k!='what'})
print('search', j.get('search'))
print('wer', j.get('wer_vs_transcripts',{}).get('wer_percent'), j.get('search_retries_last_batch'), j.get('search_grid_last_batch'))
PY
bash tools/profile_bench.sh r3b; echo "profile rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_feat.py tests/test_gpu_align.py -q -m gpu -s 2>&1 | grep -E "words equal|stage errors|GPU search behind|   [A-Za-z].* [0-9]$|CompressedMatrix|chirp \(|passed|failed" > gpurun_out/r3_gpu_tests_lines.log; tail -3 gpurun_out/r3_gpu_tests_lines.log
