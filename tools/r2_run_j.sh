set -o pipefail
python -m pytest tests -m gpu -q -x -k "steps or binfuncs" > gpurun_out/r2_t13.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2_t13.log
for th in 1 2 4; do
  VBH_DEBUG_TIMES=1 timeout -k 10 300 python tools/bench_steps.py --utts 1024 --nj 4 --fmllr --job-threads $th > gpurun_out/r2_steps_$th.json 2> gpurun_out/r2_steps_$th.err; echo "steps threads=$th rc=$? faults=$(grep -c 'Memory access fault' gpurun_out/r2_steps_$th.err)"
  python -c "
import json
o=json.loads(open('gpurun_out/r2_steps_$th.json').read().strip().split('\n')[-1])
print({k:(round(v,2) if isinstance(v,float) else v) for k,v in o.items() if k.endswith('_s') or k.startswith('Decode') or k.startswith('Score') or k in ('frames_per_s',)})"
done
grep -a "lap" gpurun_out/r2_steps_4.err | sort | uniq -c | sort -rn | head -5
grep -a "lap\|DecodeFmllr" gpurun_out/r2_steps_4.err | awk '{k=$3" "$4" "$5; if ($1=="DecodeFmllr") k=$2; t[k]+=$(NF-1); n[k]++} END {for (k in t) printf "%8.0f ms %4d x  %s\n", t[k], n[k], k}' | sort -rn | head -24
