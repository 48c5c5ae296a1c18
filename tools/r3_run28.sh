cd $GRAFT_REPO_ROOT
for st in 16 8; do
VBH_GMM_CUT_STEP=$st LD=2528 VBH_DEBUG_GMM_PACK=1 timeout -k 10 150 python tools/gmm_time.py 2>gpurun_out/gmm_time.err || tail -5 gpurun_out/gmm_time.err; grep packing gpurun_out/gmm_time.err
done
VBH_GMM_CUT_STEP=16 G=25000 LD=2528 VBH_DEBUG_GMM_PACK=1 timeout -k 10 150 python tools/gmm_time.py 2>gpurun_out/gmm_time.err || tail -5 gpurun_out/gmm_time.err; grep packing gpurun_out/gmm_time.err
python -m pytest tests/test_gpu_gmm.py -m gpu -x -q 2>&1 | tail -2
bash tools/pmc_gmm.sh r3i "WRITE_SIZE"
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r3i_summary.txt
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_r3i_*')):
    if not d[-1].isdigit(): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            for k in ('gmm_rows',):
                if k in r['Kernel_Name']:
                    acc[(k,r['Counter_Name'])]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
        for k in acc: print(d, k, acc[k], n[k])
PY
rm -rf gpurun_out/pmc_r3i_*/
