cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
VBH_PROBE_FASTER=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_faster -- python3 $R/bench.py --probe --batch 1024 --no-cpu-baseline --no-step-api > $R/gpurun_out/r3_34.log 2>&1
grep -a "FasterDecoder" $R/gpurun_out/r3_34.log
python3 - <<PY
import csv,glob
for f in glob.glob('$R/gpurun_out/prof_faster/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'faster_kernel' in r['Name'] or 'decode_kernel' in r['Name']: print(r['Name'][:40], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
rm -rf $R/gpurun_out/prof_faster
