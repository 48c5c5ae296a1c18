cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in um4 um2; do
  VBH_LIB_PATH=$R/voicebridge_amd/lib/variants/libvoicebridge_hip_$v.so timeout 400 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_ic_$v -- python3 $R/bench.py --probe --batch 256 > $R/gpurun_out/pmc_ic_$v.log 2>&1
done
