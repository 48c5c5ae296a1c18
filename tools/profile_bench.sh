# rocprofv3 passes over the default bench (run under gpurun): kernel stats (default steps), then
# FETCH_SIZE and WRITE_SIZE in passes of their own (one step).  usage: bash tools/profile_bench.sh <tag>;
# afterwards, here: python tools/summarize_prof.py <tag> r01_x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_${tag}_bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
