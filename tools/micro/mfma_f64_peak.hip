// f64 matrix-core peak of the device, measured: every wave runs a loop of independent v_mfma_f64_16x16x4_f64 chains.
// The micro-architecture guide lists no f64 MFMA row; bench.py prices fmllr_accumulate_kernel against the vendor's
// 78.6 TFLOP/s.  usage (under gpurun): hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_peak tools/micro/mfma_f64_peak.hip && /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void peak(double *out, int iters) {
  d4 acc[8];
  for (int i = 0; i < 8; i++) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  if (s == 12345.678) out[0] = s;     // keeps the chains alive
}
int main() {
  double *d; hipMalloc(&d, 8);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, iters = 20000;
  for (int wgs_per_cu : {1, 2, 4}) {
    const int grid = cus * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(peak, dim3(grid), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 /*waves*/ * iters * 8 * 2048.0;
    printf("{\"what\": \"v_mfma_f64_16x16x4_f64, 8 independent chains per wave\", \"workgroups_per_cu\": %d, \"cus\": %d, \"ms\": %.3f, \"tflops\": %.2f}\n",
           wgs_per_cu, cus, ms, flops / (ms * 1e-3) / 1e12);
  }
  return 0;
}
