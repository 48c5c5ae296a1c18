cd $GRAFT_REPO_ROOT
for cfg in "--fe-cus 0 --search-cus 0 --pipeline 3" "--fe-cus 0 --search-cus 240 --pipeline 3" "--fe-cus 0 --search-cus 248 --pipeline 3" "--fe-cus 0 --search-cus 0 --pipeline 2" "--fe-cus 0 --search-cus 240 --pipeline 4"; do
  VBH_BENCH_TIMELINE=gpurun_out/r2_tlx.npy timeout -k 10 500 python bench.py --steps 6 --warmup 1 $cfg --no-cpu-baseline --no-step-api --no-determinize > gpurun_out/r2_fe.json 2> gpurun_out/r2_fe.err; echo "[$cfg] rc=$?"
  python - <<PY
import json, numpy as np
l=[x for x in open('gpurun_out/r2_fe.json') if x.startswith('{')][-1]; j=json.loads(l)
t=np.load('gpurun_out/r2_tlx.npy'); t0=t[...,0].min()
ov=[ (t[b,:,1].max()-t[b+1,:,0].min())/1e5 for b in range(t.shape[0]-1)]
print('   value %.0f'%j['value'], 'kernel_ms %.0f'%j['roofline']['kernel_ms'], 'gmm %.0f mfcc+post %.0f'%(j['stage_ms']['gmm'], j['stage_ms']['mfcc+featpost']), 'utt ms', [round(x) for x in j['search_grid_last_batch']['utterance_ms_min_mean_max']], 'overlap of consecutive grids ms', [round(x) for x in ov], 'gmm alone %.1f'%j['roofline_gmm']['kernel_ms'] if 'roofline_gmm' in j else '')
PY
done
