cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/tcc_counters.txt
bash tools/pmc_gmm.sh r3j "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_WRITEBACK_sum TCC_WRITE_sum" "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_REQ_sum"
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r3j_summary.txt
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_r3j_*')):
    if not d[-1].isdigit(): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            for k in ('gmm_rows',):
                if k in r['Kernel_Name']:
                    acc[(k,r['Counter_Name'])]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
        for k in acc: print(d, k, acc[k], n[k])
PY
tail -3 gpurun_out/pmc_r3j_*.log
rm -rf gpurun_out/pmc_r3j_*/
