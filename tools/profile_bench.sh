# rocprofv3 passes over the default bench (run under gpurun): kernel stats (default steps), then
# FETCH_SIZE and WRITE_SIZE in passes of their own (one step).  usage: bash tools/profile_bench.sh <tag>;
# afterwards, here: python tools/summarize_prof.py <tag> r02_x
# The PCM of the workload is synthesised once by an unprofiled run (--save-waves) and read back by the profiled ones: the
# counter passes do not survive torch's own kernels (SIGSEGV in the profiler right after start-up, r02).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1
python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-step-api --no-determinize --save-waves /tmp/vb_waves.npy > $R/gpurun_out/prof_${tag}_save.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag} -- python3 $R/bench.py --no-cpu-baseline --no-step-api --load-waves /tmp/vb_waves.npy > $R/gpurun_out/prof_${tag}_bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-step-api --no-determinize --load-waves /tmp/vb_waves.npy > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
