// f32 matrix-core rate of the device, measured: every wave runs a loop of independent v_mfma_f32_32x32x2_f32 chains (the instruction
// gmm_rows_kernel and featpost_kernel issue).  MI355X_MICROARCH.md lists 157.3 TFLOP/s for it; this says what a kernel that does
// nothing else sustains on the box (clocks under matrix load included), i.e. what gmm_rows_kernel's fraction should be read against.
// usage (under gpurun): hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_peak tools/micro/mfma_f32_peak.hip && /tmp/mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void peak(float *out, int iters) {
  f16v acc[CH];
  for (int i = 0; i < CH; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CH; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.0f;
  for (int i = 0; i < CH; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;     // keeps the chains alive
}
template <int CH>
void run(float *d, int cus, int wgs_per_cu) {
  const int grid = cus * wgs_per_cu, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(peak<CH>, dim3(grid), dim3(256), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(peak<CH>, dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 /*waves*/ * iters * CH * 4096.0;      // 32 x 32 x 2 x 2 flop per instruction
  printf("{\"what\": \"v_mfma_f32_32x32x2_f32, %d independent chains per wave\", \"workgroups_per_cu\": %d, \"cus\": %d, \"ms\": %.3f, \"tflops\": %.2f}\n",
         CH, wgs_per_cu, cus, ms, flops / (ms * 1e-3) / 1e12);
}
int main() {
  float *d; hipMalloc(&d, 8);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  for (int w : {1, 2, 3}) { run<2>(d, cus, w); run<4>(d, cus, w); }
  return 0;
}
