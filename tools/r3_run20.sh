cd $GRAFT_REPO_ROOT
bash tools/pmc_gmm.sh r3g "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_r3g_*')):
    if not d[-1].isdigit(): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            if 'gmm_rows' in r['Kernel_Name']:
                acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
        for k in acc: print(d, k, acc[k], n[k])
PY
