// vbh_ctx.hip -- context, memory and stopwatch entry points of the C ABI.
#include "vbh_internal.h"

static thread_local char g_err[1024] = "";

void vbh_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

vbh_arena *vbh_arena_begin(vbh_ctx *ctx, size_t bytes) {
  vbh_arena *a = &ctx->arena[ctx->arena_next];
  ctx->arena_next = (ctx->arena_next + 1) % VBH_ARENAS;
  auto fail = [](hipError_t e, const char *what) { vbh_set_error("%s failed: %s", what, hipGetErrorString(e)); return (vbh_arena *)nullptr; };
  hipError_t e;
  if (a->pending) {
    if ((e = hipEventSynchronize(a->done)) != hipSuccess) return fail(e, "hipEventSynchronize(arena)");
    a->pending = false;
  }
  if (!a->done && (e = hipEventCreateWithFlags(&a->done, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate(arena)");
  if (bytes > a->cap) {
    // growth only: the previous buffers may still be read by kernels in flight, so drain first
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(e, "hipStreamSynchronize(arena)");
    if (a->h) (void)hipHostFree(a->h);
    if (a->d) (void)hipFree(a->d);
    a->h = a->d = nullptr; a->cap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&a->h), want, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc(arena)");
    if ((e = hipMalloc(reinterpret_cast<void **>(&a->d), want)) != hipSuccess) return fail(e, "hipMalloc(arena)");
    a->cap = want;
  }
  a->used = 0;
  return a;
}

int vbh_arena_commit(vbh_ctx *ctx, vbh_arena *a) {
  if (a->used) VBH_HIP(hipMemcpyAsync(a->d, a->h, a->used, hipMemcpyHostToDevice, ctx->stream));
  VBH_HIP(hipEventRecord(a->done, ctx->stream));
  a->pending = true;
  return 0;
}


void *vbh_scratch(vbh_ctx *c, int slot, size_t bytes) {
  if (!c || slot < 0 || slot >= VBH_SCRATCH_SLOTS) { vbh_set_error("vbh_scratch: bad slot %d", slot); return nullptr; }
  if (c->scratch[slot].cap >= bytes && c->scratch[slot].p) return c->scratch[slot].p;
  if (hipSetDevice(c->device) != hipSuccess) { vbh_set_error("vbh_scratch: hipSetDevice failed"); return nullptr; }
  if (c->scratch[slot].p) {
    (void)hipStreamSynchronize(c->stream);      // (whoever used the old buffer on this stream is done; other streams never see it)
    (void)hipFree(c->scratch[slot].p);
    c->scratch[slot].p = nullptr; c->scratch[slot].cap = 0;
  }
  const size_t want = bytes + bytes / 4 + 256;      // room to grow: a job's batches differ by a few per cent
  const hipError_t e = hipMalloc(&c->scratch[slot].p, want);
  if (e != hipSuccess) { c->scratch[slot].p = nullptr; vbh_set_error("vbh_scratch: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return nullptr; }
  c->scratch[slot].cap = want;
  return c->scratch[slot].p;
}

extern "C" {

int vbh_abi_version(void) { return VBH_ABI_VERSION; }
const char *vbh_last_error(void) { return g_err; }

int vbh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int ctx_create(int device_id, int priority, int first_cu, int n_cus, vbh_ctx **out);
int vbh_ctx_create(int device_id, vbh_ctx **out) { return ctx_create(device_id, 0, 0, 0, out); }
int vbh_ctx_create_with_priority(int device_id, int priority, vbh_ctx **out) { return ctx_create(device_id, priority, 0, 0, out); }
int vbh_ctx_create_on_cus(int device_id, int first_cu, int n_cus, vbh_ctx **out) {
  VBH_CHECK(n_cus > 0 && first_cu >= 0, "vbh_ctx_create_on_cus: first_cu = %d, n_cus = %d", first_cu, n_cus);
  return ctx_create(device_id, 0, first_cu, n_cus, out);
}
static int ctx_create(int device_id, int priority, int first_cu, int n_cus, vbh_ctx **out) {
  VBH_CHECK(out != nullptr, "vbh_ctx_create: out is NULL");
  int n = vbh_device_count();
  VBH_CHECK(n > 0, "vbh_ctx_create: no HIP device visible (this library has no CPU path)");
  VBH_CHECK(device_id >= 0 && device_id < n, "vbh_ctx_create: device %d out of range (%d devices)", device_id, n);
  VBH_HIP(hipSetDevice(device_id));
  vbh_ctx *c = new vbh_ctx();
  c->device = device_id;
  hipDeviceProp_t prop;
  VBH_HIP(hipGetDeviceProperties(&prop, device_id));
  c->num_cus = prop.multiProcessorCount;
  if (n_cus > 0 && first_cu < c->num_cus && (first_cu > 0 || n_cus < c->num_cus)) {
    // a stream confined to bits [first_cu, first_cu + n_cus) of the CU mask; the driver deals
    // consecutive bits round-robin over the XCDs, so every XCD contributes n_cus / 8 of its CUs.
    // Contexts on disjoint ranges never wait for each other's workgroups to leave a CU.
    if (first_cu + n_cus > c->num_cus) n_cus = c->num_cus - first_cu;
    uint32_t mask[32] = {0};
    for (int i = first_cu; i < first_cu + n_cus && i < 1024; i++) mask[i >> 5] |= 1u << (i & 31);
    VBH_HIP(hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)((c->num_cus + 31) / 32), mask));
    c->num_cus = n_cus;
  } else {
    // priority > 0: its kernels get CUs before those of other streams (least = 0, the default);
    // mapped onto the device's range (numerically lower = higher priority in HIP)
    int least = 0, greatest = 0;
    VBH_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    int p = least - priority;
    if (p < greatest) p = greatest;
    if (p > least) p = least;
    VBH_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, p));
  }
  for (int i = 0; i < VBH_MAX_TIMERS; i++) {
    VBH_HIP(hipEventCreate(&c->ev_start[i]));
    VBH_HIP(hipEventCreate(&c->ev_stop[i]));
  }
  *out = c;
  return 0;
}

void vbh_ctx_destroy(vbh_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < VBH_ARENAS; i++) {
    if (c->arena[i].h) (void)hipHostFree(c->arena[i].h);
    if (c->arena[i].d) (void)hipFree(c->arena[i].d);
    if (c->arena[i].done) (void)hipEventDestroy(c->arena[i].done);
  }
  for (int i = 0; i < VBH_SCRATCH_SLOTS; i++) if (c->scratch[i].p) (void)hipFree(c->scratch[i].p);
  for (int i = 0; i < VBH_MAX_TIMERS; i++) {
    if (c->marks[i]) (void)hipEventDestroy(c->marks[i]);
    (void)hipEventDestroy(c->ev_start[i]);
    (void)hipEventDestroy(c->ev_stop[i]);
  }
  (void)hipStreamDestroy(c->stream);
  delete c;
}

int vbh_ctx_sync(vbh_ctx *c) {
  if (vbh_use_device(c)) return -1;
  VBH_HIP(hipStreamSynchronize(c->stream));
  return 0;
}
int vbh_ctx_wait_for(vbh_ctx *c, vbh_ctx *other) {
  VBH_CHECK(c && other, "vbh_ctx_wait_for: NULL argument");
  VBH_CHECK(c->device == other->device, "vbh_ctx_wait_for: contexts on different devices");
  if (vbh_use_device(c)) return -1;
  if (c == other) return 0;
  hipEvent_t e;
  VBH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipError_t r = hipEventRecord(e, other->stream);
  if (r == hipSuccess) r = hipStreamWaitEvent(c->stream, e, 0);
  (void)hipEventDestroy(e);      // released once the recorded work has completed
  VBH_HIP(r);
  return 0;
}
int vbh_ctx_mark(vbh_ctx *c, int slot) {
  VBH_CHECK(c && slot >= 0 && slot < VBH_MAX_TIMERS, "vbh_ctx_mark: bad argument");
  if (vbh_use_device(c)) return -1;
  if (!c->marks[slot]) VBH_HIP(hipEventCreateWithFlags(&c->marks[slot], hipEventDisableTiming));
  VBH_HIP(hipEventRecord(c->marks[slot], c->stream));
  return 0;
}
int vbh_ctx_wait_mark(vbh_ctx *c, vbh_ctx *other, int slot) {
  VBH_CHECK(c && other && slot >= 0 && slot < VBH_MAX_TIMERS, "vbh_ctx_wait_mark: bad argument");
  VBH_CHECK(c->device == other->device, "vbh_ctx_wait_mark: contexts on different devices");
  VBH_CHECK(other->marks[slot], "vbh_ctx_wait_mark: mark %d was never set", slot);
  if (vbh_use_device(c)) return -1;
  if (c == other) return 0;
  VBH_HIP(hipStreamWaitEvent(c->stream, other->marks[slot], 0));
  return 0;
}
void *vbh_ctx_stream(vbh_ctx *c) { return reinterpret_cast<void *>(c->stream); }
int vbh_ctx_device(vbh_ctx *c) { return c->device; }

int vbh_ctx_scratch(vbh_ctx *c, int32_t slot, size_t bytes, void **d_ptr) {
  VBH_CHECK(c && d_ptr && slot >= 0 && VBH_SCR_USER + slot < VBH_SCRATCH_SLOTS, "vbh_ctx_scratch: bad argument (slots 0 .. %d)", VBH_SCRATCH_SLOTS - VBH_SCR_USER - 1);
  *d_ptr = vbh_scratch(c, VBH_SCR_USER + slot, bytes ? bytes : 1);
  return *d_ptr ? 0 : -1;
}

// Gives a scratch slot's memory back (slot < 0: every user slot's).  The slots only grow otherwise, and a host that put a job's score matrix
// in one keeps gigabytes pinned per thread for the life of its context: the host layer calls this when it lets go of its device
// cache (vb_release_device_cache), i.e. at the end of a step when asked to, or when the thread's device changes.  Waits for the context's stream.
int vbh_ctx_scratch_release(vbh_ctx *c, int32_t slot) {
  VBH_CHECK(c && VBH_SCR_USER + slot < VBH_SCRATCH_SLOTS, "vbh_ctx_scratch_release: bad argument (slots 0 .. %d, or < 0 for all)", VBH_SCRATCH_SLOTS - VBH_SCR_USER - 1);
  if (vbh_use_device(c)) return -1;
  bool synced = false;
  for (int i = VBH_SCR_USER; i < VBH_SCRATCH_SLOTS; i++) {
    if (slot >= 0 && i != VBH_SCR_USER + slot) continue;
    if (!c->scratch[i].p) continue;
    if (!synced) { VBH_HIP(hipStreamSynchronize(c->stream)); synced = true; }
    (void)hipFree(c->scratch[i].p);
    c->scratch[i].p = nullptr; c->scratch[i].cap = 0;
  }
  return 0;
}

int vbh_dev_alloc(vbh_ctx *c, size_t bytes, void **d_ptr) {
  if (vbh_use_device(c)) return -1;
  VBH_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
  return 0;
}
int vbh_dev_free(vbh_ctx *c, void *d_ptr) {
  if (vbh_use_device(c)) return -1;
  VBH_HIP(hipStreamSynchronize(c->stream));
  VBH_HIP(hipFree(d_ptr));
  return 0;
}
int vbh_copy_h2d(vbh_ctx *c, void *d_dst, const void *h_src, size_t bytes) {
  if (vbh_use_device(c)) return -1;
  if (bytes) VBH_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
  // pageable host memory: make the copy complete before the caller may reuse h_src
  VBH_HIP(hipStreamSynchronize(c->stream));
  return 0;
}
int vbh_copy_d2h(vbh_ctx *c, void *h_dst, const void *d_src, size_t bytes) {
  if (vbh_use_device(c)) return -1;
  if (bytes) VBH_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  VBH_HIP(hipStreamSynchronize(c->stream));
  return 0;
}
int vbh_memset(vbh_ctx *c, void *d_dst, int value, size_t bytes) {
  if (vbh_use_device(c)) return -1;
  if (bytes) VBH_HIP(hipMemsetAsync(d_dst, value, bytes, c->stream));
  return 0;
}

int vbh_timer_start(vbh_ctx *c, int slot) {
  VBH_CHECK(slot >= 0 && slot < VBH_MAX_TIMERS, "timer slot %d out of range", slot);
  if (vbh_use_device(c)) return -1;
  VBH_HIP(hipEventRecord(c->ev_start[slot], c->stream));
  return 0;
}
int vbh_timer_stop(vbh_ctx *c, int slot) {
  VBH_CHECK(slot >= 0 && slot < VBH_MAX_TIMERS, "timer slot %d out of range", slot);
  if (vbh_use_device(c)) return -1;
  VBH_HIP(hipEventRecord(c->ev_stop[slot], c->stream));
  return 0;
}
int vbh_timer_elapsed_ms(vbh_ctx *c, int slot, float *ms) {
  VBH_CHECK(slot >= 0 && slot < VBH_MAX_TIMERS, "timer slot %d out of range", slot);
  if (vbh_use_device(c)) return -1;
  VBH_HIP(hipEventSynchronize(c->ev_stop[slot]));
  VBH_HIP(hipEventElapsedTime(ms, c->ev_start[slot], c->ev_stop[slot]));
  return 0;
}

}  // extern "C"
