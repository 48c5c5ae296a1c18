# r4: same-box search-only A/B (bench.py --probe --batch 1024): usage bash tools/probe_ab.sh name[:admission] ...   (name = product | variant)
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  v=${spec%%:*}; a=${spec#*:}; [ "$a" = "$spec" ] && a=ordered
  if [ "$v" = product ]; then unset VBH_LIB_PATH; else export VBH_LIB_PATH=$GRAFT_REPO_ROOT/voicebridge_amd/lib/variants/libvoicebridge_hip_$v.so; fi
  VBH_ADMISSION=$a timeout -k 10 300 python bench.py --probe --batch 1024 --no-cpu-baseline > gpurun_out/probe_${v}_$a.log 2>&1; echo "$spec rc=$?"
  grep -h "decode of\|all utterances\|phase share" gpurun_out/probe_${v}_$a.log | head -4
done
