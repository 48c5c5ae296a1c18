# r4: DecodeFmllr through the step API (tools/bench_steps.py --utts 1024 --fmllr) with search grids capped at the given workgroup counts
cd $GRAFT_REPO_ROOT
for w in "$@"; do
  VBH_SEARCH_MAX_WGS=$w timeout -k 10 400 python tools/bench_steps.py --utts 1024 --fmllr > gpurun_out/steps_w$w.log 2>&1; echo "wgs=$w rc=$?"
  tail -1 gpurun_out/steps_w$w.log | python3 -c "
import json,sys
j=json.loads(sys.stdin.read())
print({k.split(' ')[0]: (round(v,3) if isinstance(v,float) else v) for k,v in j.items() if k.split(' ')[0] in ('MakeMfcc_s','Decode_s','DecodeFmllr_s','ScoreKaldiWER_s','Decode_best_wer','DecodeFmllr_best_wer')})"
done
