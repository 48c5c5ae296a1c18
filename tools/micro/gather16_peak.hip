// A roof that fits the search kernels ("next" 2a): what MI355X delivers to a kernel that does nothing but gather random
// 16-byte records -- the access decode_kernel / faster_kernel are made of (a token's state record in the packed graph, its arcs behind
// it, the destination token of a link, a second-level map entry).  256 workgroups x 16 waves (one 1024-thread workgroup per CU, the
// search kernels' shape; fewer waves for comparison), every lane loading records at hashed positions of a table the size of the packed
// graph (126 MB: 7.9 M records), with
//   * ILP = 1 / 2 / 4 / 8 independent loads in flight per lane,
//   * `span` adjacent lanes sharing one 64-byte block (span 1: 64 lines per wave-instruction; span 4: a state record + three arcs),
//   * a dependent second hop (the record read names the next one: token -> state -> arcs chains), again with 1 / 2 / 4 chains per lane.
// Reported per configuration: records/s chip-wide, useful GB/s (16 B per record), 64-byte lines/s per CU.  bench.py prices the search
// kernel's random-access lines per second against the best independent figure of this table (roofline.gather_frac).
// usage (under gpurun): hipcc --offload-arch=gfx950 -O3 -o /tmp/gather16_peak tools/micro/gather16_peak.hip && /tmp/gather16_peak [table MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ void fill(i32x4 *t, unsigned n) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    i32x4 r; r.x = (int)(mix(i * 2654435761u + 1u) % n); r.y = (int)(mix(i + 0x9e3779b9u) % n); r.z = (int)i; r.w = 0;
    t[i] = r;
  }
}

// independent gathers: ILP loads per lane and iteration, addresses from a hash of (thread, iteration, k); groups of SPAN lanes read SPAN
// consecutive records of one 64-byte-aligned block
template <int ILP, int SPAN>
__global__ __launch_bounds__(1024) void gather_indep(const i32x4 *__restrict__ t, unsigned n, int iters, int *out) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned grp = gid / SPAN, sub = gid % SPAN;
  const unsigned nblk = n / 4;      // 64-byte blocks
  int acc = 0;
  for (int it = 0; it < iters; it++) {
    i32x4 r[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) {
      const unsigned b = mix(grp * 0x9e3779b9u + (unsigned)(it * ILP + k) * 0x85ebca6bu) % nblk;
      r[k] = t[(size_t)b * 4 + (SPAN == 1 ? (gid & 3u) : sub)];
    }
#pragma unroll
    for (int k = 0; k < ILP; k++) acc ^= r[k].x + r[k].w;
  }
  if (acc == 0x12345678) out[0] = acc;
}

// dependent chains: CH chains per lane, every hop's address is made from the previous record's .x (a uniformly random record) and the
// hop number -- iterating one fixed function would run into its short cycles (~sqrt(n) records, cache resident) after a few hundred hops
template <int CH>
__global__ __launch_bounds__(1024) void gather_chain(const i32x4 *__restrict__ t, unsigned n, int hops, int *out) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned p[CH];
#pragma unroll
  for (int k = 0; k < CH; k++) p[k] = mix(gid * 0x9e3779b9u + k) % n;
  int acc = 0;
  for (int h = 0; h < hops; h++) {
    i32x4 r[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) r[k] = t[p[k]];
#pragma unroll
    for (int k = 0; k < CH; k++) { p[k] = mix((unsigned)r[k].x + (unsigned)h * 0x9e3779b9u) % n; acc ^= r[k].z; }
  }
  if (acc == 0x12345678) out[0] = acc;
}

// two-hop form of a search phase: a fresh random record (the token's state), then the record it names (its arcs): ILP pairs per lane
template <int ILP>
__global__ __launch_bounds__(1024) void gather_two_hop(const i32x4 *__restrict__ t, unsigned n, int iters, int *out) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  int acc = 0;
  for (int it = 0; it < iters; it++) {
    i32x4 a[ILP], b[ILP];
#pragma unroll
    for (int k = 0; k < ILP; k++) a[k] = t[mix(gid * 0x9e3779b9u + (unsigned)(it * ILP + k) * 0x85ebca6bu) % n];
#pragma unroll
    for (int k = 0; k < ILP; k++) b[k] = t[(unsigned)a[k].y];
#pragma unroll
    for (int k = 0; k < ILP; k++) acc ^= b[k].z;
  }
  if (acc == 0x12345678) out[0] = acc;
}

// random 16-byte stores (a link / token record written to a hashed position): what the write side of such a kernel can do
template <int ILP>
__global__ __launch_bounds__(1024) void scatter_indep(i32x4 *t, unsigned n, int iters) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) {
      const unsigned i = mix(gid * 0x9e3779b9u + (unsigned)(it * ILP + k) * 0x85ebca6bu) % n;
      i32x4 r; r.x = (int)(mix(i * 2654435761u + 1u) % n); r.y = (int)(mix(i + 0x9e3779b9u) % n); r.z = (int)i; r.w = it;      // (the table stays a table of indices)
      t[i] = r;
    }
  }
}

struct Cfg { int cus; i32x4 *t; unsigned n; int *out; double table_mb; };

template <class L>
static void timed(const Cfg &c, const char *what, int threads, double records_per_thread, int lines_per_record_x4, L launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(true);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    launch(false);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double recs = (double)c.cus * threads * records_per_thread;
  const double rps = recs / (best * 1e-3);
  const double lines = recs * lines_per_record_x4 / 4.0;
  printf("{\"what\": \"%s\", \"cus\": %d, \"waves_per_cu\": %d, \"table_mb\": %.0f, \"ms\": %.3f, \"records_per_s\": %.4g, \"useful_gb_per_s\": %.1f, "
         "\"lines64_per_s_per_cu\": %.4g, \"line_gb_per_s\": %.1f}\n",
         what, c.cus, threads / 64, c.table_mb, best, rps, rps * 16.0 / 1e9, lines / (best * 1e-3) / c.cus, lines / (best * 1e-3) * 64.0 / 1e9);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const double mb = argc > 1 ? atof(argv[1]) : 126.0;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  Cfg c; c.cus = p.multiProcessorCount; c.table_mb = mb;
  c.n = (unsigned)(mb * 1e6 / 16.0) & ~3u;
  hipMalloc(&c.t, (size_t)c.n * 16); hipMalloc(&c.out, 64);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, c.t, c.n);
  hipDeviceSynchronize();
  const int IT = 2048;
#define INDEP(ILP, SPAN, TH)                                                                                                     \
  timed(c, "independent, " #ILP " in flight per lane, " #SPAN " lanes per 64-byte block", TH, (double)IT * ILP / ((ILP) >= 4 ? 2 : 1), \
        SPAN == 1 ? 4 : 4 / SPAN, [&](bool warm) {                                                                               \
          hipLaunchKernelGGL((gather_indep<ILP, SPAN>), dim3(c.cus), dim3(TH), 0, 0, c.t, c.n, warm ? 8 : IT / ((ILP) >= 4 ? 2 : 1), c.out); \
        });
  for (int th : {1024, 512, 256}) {
    if (th == 1024) { INDEP(1, 1, 1024) INDEP(2, 1, 1024) INDEP(4, 1, 1024) INDEP(8, 1, 1024) INDEP(1, 4, 1024) INDEP(2, 4, 1024) INDEP(4, 4, 1024) }
    if (th == 512) { INDEP(1, 1, 512) INDEP(4, 1, 512) INDEP(8, 1, 512) }
    if (th == 256) { INDEP(1, 1, 256) INDEP(4, 1, 256) INDEP(8, 1, 256) }
  }
#define CHAIN(CH, TH)                                                                                                             \
  timed(c, "dependent chain, " #CH " chains per lane", TH, (double)(IT / 2) * CH, 4, [&](bool warm) {                             \
    hipLaunchKernelGGL((gather_chain<CH>), dim3(c.cus), dim3(TH), 0, 0, c.t, c.n, warm ? 8 : IT / 2, c.out);                      \
  });
  CHAIN(1, 1024) CHAIN(2, 1024) CHAIN(4, 1024) CHAIN(1, 256)
#define TWOHOP(ILP, TH)                                                                                                           \
  timed(c, "two hops (random record, then the record it names), " #ILP " pairs in flight per lane", TH, (double)(IT / 2) * ILP * 2, 4, [&](bool warm) { \
    hipLaunchKernelGGL((gather_two_hop<ILP>), dim3(c.cus), dim3(TH), 0, 0, c.t, c.n, warm ? 8 : IT / 2, c.out);                   \
  });
  TWOHOP(1, 1024) TWOHOP(2, 1024) TWOHOP(4, 1024)
  // the same gathers with part of the chip idle: is the ceiling the CU's (its L1's miss queue) or the fabric's?
  const int all = c.cus;
  for (int cus : {128, 64, 32, 8}) {
    c.cus = cus;
    INDEP(1, 1, 1024) INDEP(4, 1, 1024) INDEP(4, 1, 256) TWOHOP(2, 1024)
  }
  c.cus = all;
  timed(c, "random 16-byte stores, 1 per lane and iteration", 1024, (double)IT, 4, [&](bool warm) {
    hipLaunchKernelGGL((scatter_indep<1>), dim3(c.cus), dim3(1024), 0, 0, c.t, c.n, warm ? 8 : IT); });
  timed(c, "random 16-byte stores, 4 per lane and iteration", 1024, (double)IT * 2, 4, [&](bool warm) {
    hipLaunchKernelGGL((scatter_indep<4>), dim3(c.cus), dim3(1024), 0, 0, c.t, c.n, warm ? 8 : IT / 2); });
  return 0;
}
