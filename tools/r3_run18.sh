cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_peak tools/micro/mfma_f32_peak.hip && /tmp/mfma_f32_peak | tee gpurun_out/r03_mfma_f32_peak.json
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_peak tools/micro/mfma_f64_peak.hip && /tmp/mfma_f64_peak
