set -o pipefail
python -m pytest tests -m gpu -q -x > gpurun_out/r2_t12.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2_t12.log
for th in 1 2 4; do
  timeout -k 10 300 python tools/bench_steps.py --utts 1024 --nj 4 --fmllr --job-threads $th > gpurun_out/r2_steps_$th.json 2> gpurun_out/r2_steps_$th.err; echo "steps threads=$th rc=$? faults=$(grep -c 'Memory access fault' gpurun_out/r2_steps_$th.err)"
  python -c "
import json
o=json.loads(open('gpurun_out/r2_steps_$th.json').read().strip().split('\n')[-1])
print({k:(round(v,2) if isinstance(v,float) else v) for k,v in o.items() if k.endswith('_s') or k.startswith('Decode') or k.startswith('Score') or k in ('frames_per_s',)})"
done
