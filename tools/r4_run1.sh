# r4: ordered admission -- parity tests of the search, then a same-box A/B of the two admission rules (4 steps each)
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_config3.py tests/test_gpu_gmm.py -x -q -m gpu > gpurun_out/r4_t1.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4_t1.log
for a in exact ordered; do
  VBH_ADMISSION=$a timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-step-api --no-determinize > gpurun_out/r4_ab_$a.json 2> gpurun_out/r4_ab_$a.err; echo "$a rc=$?"
  python - <<PY
import json
l=[x for x in open('gpurun_out/r4_ab_$a.json') if x.startswith('{')][-1]; j=json.loads(l)
print('$a', 'value %.0f'%j['value'], 'kernel_ms %.1f'%j['roofline']['kernel_ms'], j['search_grid_last_batch']['utterance_ms_min_mean_max'], j['search_retries_last_batch'], j.get('wer_vs_transcripts'))
PY
done
