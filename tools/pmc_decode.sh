# PMC passes over one decode of BATCH utterances (bench.py --probe); one counter set per pass.
# usage (under gpurun): bash tools/pmc_decode.sh <tag> <batch> "<set1>" "<set2>" ...
cd /tmp && export TMPDIR=/tmp
export VBH_BENCH_SOFT_EXIT=1
R=$GRAFT_REPO_ROOT
tag=$1; batch=$2; shift 2
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/bench.py --probe --batch $batch --no-cpu-baseline --no-step-api > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float)
for f in glob.glob('$R/gpurun_out/pmc_${tag}_$i/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'decode_kernel' in r['Kernel_Name']: tot[r['Counter_Name']]+=float(r['Counter_Value'])
print('$set ->', dict(tot))
PY
done
