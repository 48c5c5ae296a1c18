// vbh_internal.h -- shared internals of libvoicebridge_hip.so (not installed).
#ifndef VBH_INTERNAL_H_
#define VBH_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "voicebridge_hip.h"

#define VBH_MAX_TIMERS 16

// Per-call tables (offsets, chunk lists, per-speaker transforms ...) go to the device through a
// small ring of staging arenas: the host fills a pinned buffer, ONE asynchronous copy moves it to
// the arena's device buffer, kernels of that call read it there.  No allocation, no blocking copy
// and no stream synchronisation per call, so stages enqueue without waiting for the GPU -- which
// is what lets batches overlap on several contexts.  An arena is reused four calls later, after
// waiting for its own copy (long done by then).
#define VBH_ARENAS 4
struct vbh_arena {
  char *h = nullptr, *d = nullptr;
  size_t cap = 0, used = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
};

// Scratch device memory of a context: per-call work buffers (statistics, posteriors, front-end output) come from a small table of
// grow-only allocations instead of a hipMalloc / hipFree pair per call.  hipFree waits for the WHOLE device, i.e. for the persistent search
// grids of every other context: with four job threads sharing a GPU a call that frees a dozen temporaries spent most of its time waiting
// for other jobs' searches (r04: gmm-est-fmllr 134 ms alone, 400-650 ms beside searches).  A slot is reused by the next call
// that asks for it (same stream: ordered); all of it is released with the context.
#define VBH_SCRATCH_SLOTS 32
enum { VBH_SCR_FMLLR = 0, VBH_SCR_CMVN = 16, VBH_SCR_USER = 20 };      // first slot of each user (vbh_ctx_scratch serves VBH_SCR_USER ...)
struct vbh_ctx {
  int device;
  hipStream_t stream;
  hipEvent_t ev_start[VBH_MAX_TIMERS], ev_stop[VBH_MAX_TIMERS];
  int num_cus;
  vbh_arena arena[VBH_ARENAS];
  int arena_next = 0;
  hipEvent_t marks[VBH_MAX_TIMERS] = {};      // vbh_ctx_mark / vbh_ctx_wait_mark
  struct { void *p; size_t cap; } scratch[VBH_SCRATCH_SLOTS] = {};
  int fmllr_dim = 0, fmllr_nspk = 0;       // the statistics the last vbh_fmllr_accumulate* left in the scratch slots (vbh_fmllr_update_full reads them there)
};
// slot's buffer, at least `bytes` long (nullptr + error on failure); contents are not preserved when it grows
void *vbh_scratch(vbh_ctx *ctx, int slot, size_t bytes);

vbh_arena *vbh_arena_begin(vbh_ctx *ctx, size_t bytes);          // NULL on error
int vbh_arena_commit(vbh_ctx *ctx, vbh_arena *a);                 // enqueue the copy on ctx->stream
// reserve n elements in the arena, copy src there (src may be NULL: zero fill), return the device address
template <class T>
static inline T *vbh_arena_put(vbh_arena *a, const T *src, size_t n) {
  const size_t off = (a->used + 255) & ~(size_t)255, bytes = n * sizeof(T);
  if (off + bytes > a->cap) return nullptr;       // cannot happen: begin() sized the arena
  if (src) memcpy(a->h + off, src, bytes); else memset(a->h + off, 0, bytes);
  a->used = off + bytes;
  return reinterpret_cast<T *>(a->d + off);
}
static inline size_t vbh_arena_need(size_t bytes) { return ((bytes + 255) & ~(size_t)255) + 256; }

void vbh_set_error(const char *fmt, ...);

#define VBH_HIP(call)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      vbh_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -1;                                                                         \
    }                                                                                    \
  } while (0)

#define VBH_CHECK(cond, ...)              \
  do {                                    \
    if (!(cond)) {                        \
      vbh_set_error(__VA_ARGS__);         \
      return -1;                          \
    }                                     \
  } while (0)

static inline int vbh_use_device(const vbh_ctx *ctx) {
  VBH_HIP(hipSetDevice(ctx->device));
  return 0;
}

// small RAII device buffer for tables uploaded at create time
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    free();
    n = count;
    if (count == 0) return 0;
    VBH_HIP(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
    if (getenv("VBH_DEBUG_ALLOC")) fprintf(stderr, "[vbh alloc] DevBuf [%p, %p) %zu B\n", (void *)p, (void *)((char *)p + count * sizeof(T)), count * sizeof(T));
    return 0;
  }
  int upload(const T *h, size_t count, hipStream_t s) {
    if (alloc(count)) return -1;
    if (count) VBH_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    return 0;
  }
  void free() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { free(); }
};

// the same interface over a scratch slot of the context (nothing is freed when it goes out of scope)
template <class T>
struct ScratchBuf {
  vbh_ctx *ctx; int slot;
  T *p = nullptr;
  size_t n = 0;
  ScratchBuf(vbh_ctx *c, int s) : ctx(c), slot(s) {}
  int alloc(size_t count) {
    n = count;
    p = static_cast<T *>(vbh_scratch(ctx, slot, (count ? count : 1) * sizeof(T)));
    return p ? 0 : -1;
  }
  int upload(const T *h, size_t count, hipStream_t s) {
    if (alloc(count)) return -1;
    if (count) VBH_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    return 0;
  }
};

// what another translation unit needs of a vbh_graph (defined in vbh_decode.hip): the device
// records described there (state {first arc, #eps arcs, #arcs, final}, arc {next | has-eps flag,
// weight, pdf, olabel}, eps arcs first within a state) and the ilabels
struct vbh_graph_view {
  int S, start; long long A; int max_ilabel, max_degree;
  // the device graph (vbh_decode.hip, DevGraph): one array of 16-byte records, a state's record {the graph's own id, #epsilon arcs,
  // #arcs, final cost} directly in front of its arcs, a state's id = the index of its record; pk_il: an arc's ilabel (-1: not an
  // arc); h_pos[s] (host) = state s's record
  long long PKN; int start_pk; const void *pk; const int *pk_il; const int *h_pos;
};
void vbh_graph_view_get(const vbh_graph *g, vbh_graph_view *v);
// make the arcs' pdf fields hold h_t2p[ilabel] (no-op when they already do); h_t2p[0] is ignored
int vbh_graph_fill_pdf(const vbh_graph *g, hipStream_t st, const int32_t *h_t2p, int32_t num_tids);
// every successful vbh_graph_fill_pdf is paired with one release once the search that reads the pdf fields has finished
void vbh_graph_release_pdf(const vbh_graph *g);

// what another translation unit needs of a vbh_gmm (defined in vbh_gmm.hip): row-major model arrays on the device
struct vbh_gmm_view {
  int D, P, G;
  const int *pdf_offs; const float *gconsts, *miv, *iv;
  vbh_ctx *ctx;
};
void vbh_gmm_view_get(const vbh_gmm *g, vbh_gmm_view *v);

#endif  // VBH_INTERNAL_H_
