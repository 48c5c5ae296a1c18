# Registers, spills and LDS of the search kernels as hipcc builds them (here, no GPU needed; ~10 s per configuration):
#   bash tools/kres.sh [extra -D flags]
# product = layout 1 (1024 threads per utterance); nt512 = layout 2 (two workgroups per CU, light searches), whose LDS must stay <= 81920 bytes
# (voicebridge_amd/build.py LAYOUT2_FLAGS) -- a 16-byte overshoot silently halves its occupancy: the light searches lose their gain and the
# race-detector test (tests/test_gpu_variant.py) tests nothing.
R=$(cd $(dirname $0)/.. && pwd)
F="--offload-arch=gfx950 -O3 -std=c++17 -I$R/include -I$R/voicebridge_amd/csrc -ffp-contract=off --cuda-device-only --no-gpu-bundle-output"
show() { /opt/rocm/lib/llvm/bin/llvm-readelf --notes $1 | grep -E "\.name:|\.vgpr_count|vgpr_spill|group_segment_fixed|private_segment_fixed|\.sgpr_count" | paste - - - - - - | grep -E "decode_kernel|faster_kernel" | sed -E 's/ +/ /g; s/_ZN12_GLOBAL__N_1[0-9]+//; s/EEEv.*PoolE|EvNS_6FGraph.*i//'; }
/opt/rocm/bin/hipcc $F "$@" -c $R/voicebridge_amd/csrc/vbh_decode.hip -o /tmp/kres_dec.co 2>/dev/null && echo "decode (product):" && show /tmp/kres_dec.co
/opt/rocm/bin/hipcc $F -DVBH_LAYOUT2=1 -DVBH_DECODE_NT=512 -DVBH_WAVES_PER_EU=4 -DVBH_HBITS=13 "$@" -c $R/voicebridge_amd/csrc/vbh_decode.hip -o /tmp/kres_dec512.co 2>/dev/null && echo "decode (nt512):" && show /tmp/kres_dec512.co
/opt/rocm/bin/hipcc $F "$@" -c $R/voicebridge_amd/csrc/vbh_faster.hip -o /tmp/kres_fast.co 2>/dev/null && echo "faster:" && show /tmp/kres_fast.co
