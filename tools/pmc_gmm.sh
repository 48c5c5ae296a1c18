# PMC passes over the front end only (bench.py --no-decode): one counter set per pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/bench.py --no-decode --batch 256 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_$i.log 2>&1
done
