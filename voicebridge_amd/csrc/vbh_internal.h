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

struct vbh_ctx {
  int device;
  hipStream_t stream;
  hipEvent_t ev_start[VBH_MAX_TIMERS], ev_stop[VBH_MAX_TIMERS];
  int num_cus;
};

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

#endif  // VBH_INTERNAL_H_
