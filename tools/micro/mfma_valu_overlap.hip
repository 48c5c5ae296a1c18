// Does VALU work hide under MFMAs on gfx950?  Every wave alternates a burst of NM independent v_mfma_f32_32x32x2_f32 (4 chains) with a
// burst of NV dependent-free v_fma_f32 on other registers -- (a) in the same loop body, bursts back to back (what a wave of
// gmm_rows_kernel does: 80 MFMAs, then its epilogue), (b) finely interleaved (one FMA group between MFMAs).  Reported: the MFMA rate.
// If VALU issue were free under the matrix pipe, (a) and (b) would stay near the pure-MFMA rate with three waves per SIMD.
// usage (under gpurun): hipcc --offload-arch=gfx950 -O3 -o /tmp/ov tools/micro/mfma_valu_overlap.hip 2>/dev/null && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NV, bool FINE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  f16v acc[4];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;
  float v[8];
  for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 1e-3f + i;
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; it++) {
    if (FINE) {
#pragma unroll
      for (int g = 0; g < 20; g++) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV / 20; q++) v[q & 7] = __builtin_fmaf(v[q & 7], a, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 20; g++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NV; q++) v[q & 7] = __builtin_fmaf(v[q & 7], a, b);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 4; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
  for (int i = 0; i < 8; i++) s += v[i];
  if (s == 12345.678f) out[0] = s;
}
template <int NV, bool FINE>
void run(float *d, int cus) {
  const int grid = cus * 3, iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, FINE>), dim3(grid), dim3(256), 0, 0, d, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, FINE>), dim3(grid), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 80 * 4096.0;
  printf("{\"mfma_per_burst\": 80, \"valu_fma_per_burst\": %d, \"interleaved\": %s, \"workgroups_per_cu\": 3, \"ms\": %.3f, \"mfma_tflops\": %.2f}\n", NV, FINE ? "true" : "false", ms,
         flops / (ms * 1e-3) / 1e12);
}
int main() {
  float *d; (void)hipMalloc(&d, 8);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  run<0, false>(d, cus); run<200, false>(d, cus); run<400, false>(d, cus); run<800, false>(d, cus);
  run<200, true>(d, cus); run<400, true>(d, cus); run<800, true>(d, cus);
  return 0;
}
