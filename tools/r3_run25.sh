cd $GRAFT_REPO_ROOT
bash tools/pmc_gmm.sh r3h "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r3h_summary.txt
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_r3h_*')):
    if not d[-1].isdigit(): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            for k in ('gmm_rows','gmm_loglikes_pairs'):
                if k in r['Kernel_Name']:
                    acc[(k,r['Counter_Name'])]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
        for k in acc: print(d, k, acc[k], n[k])
PY
rm -rf gpurun_out/pmc_r3h_*/
