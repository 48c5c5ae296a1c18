cd $GRAFT_REPO_ROOT
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r3_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r3_gputests.log
bash tools/profile_bench.sh r3a; echo "profile rc=$?"; tail -c 400 gpurun_out/prof_r3a_bench.log
bash tools/pmc_decode.sh r3a 256 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" 2>&1 | tail -4
