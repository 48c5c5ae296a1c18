# PMC passes over one decode of BATCH utterances (bench.py --probe); one counter set per pass.
# usage (under gpurun): bash tools/pmc_decode.sh <tag> <batch> "<set1>" "<set2>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; batch=$2; shift 2
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/bench.py --probe --batch $batch > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
