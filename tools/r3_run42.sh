cd $GRAFT_REPO_ROOT
for m in 0 1 0 1; do
  if [ $m = 1 ]; then export VB_STEP_PHASED=1; else unset VB_STEP_PHASED; fi
  timeout -k 10 600 python tools/bench_steps.py --utts 1024 --nj 4 --fmllr 2>gpurun_out/r3_42_$m.err | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('phased=$m', {k.split(' ')[0]:round(v,3) for k,v in d.items() if k.startswith('Decode') and k.endswith(')') or k.startswith('DecodeFmllr_s')}, d['DecodeFmllr_best_wer'])"
done
