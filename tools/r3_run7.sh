cd $GRAFT_REPO_ROOT
V=$PWD/voicebridge_amd/lib/variants
timeout -k 10 1200 python -m pytest tests -x -q -m gpu -s > gpurun_out/r3_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r3_gputests.log; grep -E "chirp \(|CompressedMatrix bytes|stage errors|words equal on" gpurun_out/r3_gputests.log
timeout -k 10 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-step-api > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; echo "bench rc=$? (normal exit)"; grep -c "fault" gpurun_out/r3_bench.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r3_bench.json') if l.startswith('{')][-1])
print('value', j['value'], 'raw', j.get('raw_lattice_rate',{}).get('value'))
print('gmm', {k:j['roofline_gmm'].get(k) for k in ('frac','kernel_ms','achieved')})
PY
# the round-2 fault hypothesis: two workgroups per CU on CU-MASKED search streams (round 2's default when it faulted)
for i in 1 2; do
  VBH_LIB_PATH=$V/libvoicebridge_hip_nt512.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-determinize --no-step-api --search-cus 224 > gpurun_out/r3_mask$i.json 2> gpurun_out/r3_mask$i.err; echo "nt512 masked run $i rc=$? faults=$(grep -c 'Memory access fault' gpurun_out/r3_mask$i.err)"
done
