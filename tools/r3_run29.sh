cd $GRAFT_REPO_ROOT
bash tools/pmc_gmm.sh r3i "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r3i_summary.txt
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_r3i_*')):
    if not d[-1].isdigit(): continue
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        rows=list(csv.DictReader(open(f)))
        print(list(rows[0].keys()))
        for r in rows:
            if 'gmm_rows' in r['Kernel_Name']:
                print(r.get('Dispatch_Id'), r.get('Grid_Size'), r.get('Workgroup_Size'), r['Counter_Name'], r['Counter_Value'])
PY
rm -rf gpurun_out/pmc_r3i_*/
