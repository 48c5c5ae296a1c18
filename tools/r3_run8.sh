cd $GRAFT_REPO_ROOT
V=$PWD/voicebridge_amd/lib/variants
bash tools/ab_bench.sh product um8 ul2 product
# the round-2 fault hypothesis: two workgroups per CU on CU-MASKED search streams (round 2's default when it faulted)
for i in 1 2 3; do
  VBH_LIB_PATH=$V/libvoicebridge_hip_nt512.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-determinize --no-step-api --search-cus 224 > gpurun_out/r3_mask$i.json 2> gpurun_out/r3_mask$i.err; rc=$?; echo "nt512 masked run $i rc=$rc faults=$(grep -c 'Memory access fault' gpurun_out/r3_mask$i.err)"; tail -2 gpurun_out/r3_mask$i.err | cut -c1-300
  if [ $rc = 124 ] || [ $rc = 137 ]; then break; fi
done
