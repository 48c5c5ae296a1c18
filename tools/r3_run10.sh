cd $GRAFT_REPO_ROOT
V=$PWD/voicebridge_amd/lib/variants
for v in nt512 nt512h14 nt1024h13; do
  echo "==== $v"
  VBH_LIB_PATH=$V/libvoicebridge_hip_$v.so VBH_PROBE_REPEAT=4 timeout -k 10 600 python bench.py --probe --batch 1024 --no-cpu-baseline 2>&1 | grep -E "repeat|determinism|search grid|all utterances" | cut -c1-300
done
