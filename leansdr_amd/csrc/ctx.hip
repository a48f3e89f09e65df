// leansdr_amd/csrc/ctx.hip — context, device memory and stream timing of the C ABI.
#include <cstring>
#include "lsdr_internal.h"
#include <malloc.h>
#include <mutex>

static thread_local char g_err[512] = "";

void lsdr_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int lsdr_hip_fail(hipError_t e, const char *what, const char *file, int line) {
  lsdr_set_error("%s:%d: %s failed: %s", file, line, what, hipGetErrorString(e));
  return LSDR_E_HIP;
}

extern "C" {

int lsdr_abi_version(void) { return LSDR_ABI_VERSION; }
const char *lsdr_last_error(void) { return g_err; }

int lsdr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// "0000:c1:00.0"-style PCI address of a device (for the host side to find the GPU's NUMA node under /sys/bus/pci/devices/)
int lsdr_device_pci_bus_id(int device, char *buf, int len) {
  LSDR_ARG(buf && len >= 16);
  LSDR_HIP(hipDeviceGetPCIBusId(buf, len, device));
  return LSDR_OK;
}

static int ctx_create(int device, void *hip_stream, const uint32_t *cu_mask, unsigned mask_words, lsdr_ctx **out);

int lsdr_ctx_create(int device, void *hip_stream, lsdr_ctx **out) { return ctx_create(device, hip_stream, nullptr, 0, out); }

int lsdr_ctx_create_masked(int device, const uint32_t *cu_mask, unsigned mask_words, lsdr_ctx **out) {
  LSDR_ARG(cu_mask && mask_words);
  return ctx_create(device, nullptr, cu_mask, mask_words, out);
}

static int ctx_create(int device, void *hip_stream, const uint32_t *cu_mask, unsigned mask_words, lsdr_ctx **out) {
  LSDR_ARG(out != nullptr);
  int ndev = 0;
  LSDR_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) {
    lsdr_set_error("lsdr_ctx_create: no HIP device %d (found %d) - this library has no CPU fallback", device, ndev);
    return LSDR_E_HIP;
  }
  LSDR_HIP(hipSetDevice(device));
  {
    // Host memory must not go back to the kernel while GPU queues are live.  The blocks build per-call job lists and fetch per-call
    // totals in std::vectors of hundreds of KB to MB; glibc returns such blocks with munmap / a heap trim when they are freed, the
    // unmap runs the driver's MMU notifier, and the driver suspends and restores every queue of the process: the NEXT submission,
    // whatever it is, waits ≈ 25 ms (round 2 met this as "mpeg_sync takes 24.7 ms" and moved the transfers into a pinned arena;
    // round 4 found the same stall behind viterbi_sync's "cliff" — 8PSK 2/3, 32 Mi symbols per call: 25.3 ms per call, 4.0 ms
    // under rocprofv3, 4.5 ms with MALLOC_TRIM_THRESHOLD_ / MALLOC_MMAP_THRESHOLD_ raised — although none of those vectors is ever
    // touched by a transfer).  So the first context of a process tells malloc to keep what it has: blocks up to 32 MB come from the
    // heap, and the heap is not trimmed.  LSDR_KEEP_MALLOC=1 leaves the allocator alone.
    // Contexts are created from many threads (one per capture): once per process, race-free.  Documented in include/lsdr_hip.h.
    static std::once_flag once;
    std::call_once(once, [] {
      if (!getenv("LSDR_KEEP_MALLOC")) {
        (void)mallopt(M_MMAP_THRESHOLD, 32 * 1024 * 1024);
        (void)mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
      }
    });
  }
  lsdr_ctx *c = new lsdr_ctx();
  c->device = device;
  c->own_stream = (hip_stream == nullptr);
  // any HIP failure below: release what exists so far (the context is not handed out)
#define CTX_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { \
    if (c->ev0) (void)hipEventDestroy(c->ev0); if (c->ev1) (void)hipEventDestroy(c->ev1); \
    if (c->stream && (c->own_stream || cu_mask)) (void)hipStreamDestroy(c->stream); \
    delete c; return lsdr_hip_fail(e__, #call, __FILE__, __LINE__); } } while (0)
  if (cu_mask) {
    CTX_HIP(hipExtStreamCreateWithCUMask(&c->stream, mask_words, cu_mask));
  } else if (c->own_stream) {
    const char *pe = getenv("LSDR_STREAM_PRIORITY");   // tuning hook: "high" → highest stream priority
    if (pe && !strcmp(pe, "high")) {
      int lo = 0, hi = 0;
      CTX_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      CTX_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
    } else CTX_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  }
  else c->stream = (hipStream_t)hip_stream;
  CTX_HIP(hipEventCreate(&c->ev0));
  CTX_HIP(hipEventCreate(&c->ev1));
  hipDeviceProp_t prop;
  CTX_HIP(hipGetDeviceProperties(&prop, device));
#undef CTX_HIP
  c->num_cu = prop.multiProcessorCount;
  if (cu_mask) {   // the kernels size their grids for the compute units this stream may use
    int n = 0;
    for (unsigned w = 0; w < mask_words; ++w) n += __builtin_popcount(cu_mask[w]);
    if (n > 0 && n < c->num_cu) c->num_cu = n;
  }
  c->copy_ready = false;
  *out = c;
  return LSDR_OK;
}

void lsdr_ctx_destroy(lsdr_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipFree(c->bounce);
  (void)hipFree(c->rs_tables); (void)hipFree(c->rs_counter);
  if (c->stage_base) (void)hipHostFree(c->stage_base);
  for (char *r : c->stage_retired) (void)hipHostFree(r);
  if (c->copy_ready) {
    (void)hipStreamSynchronize(c->up); (void)hipStreamSynchronize(c->down);
    (void)hipStreamDestroy(c->up); (void)hipStreamDestroy(c->down);
    (void)hipEventDestroy(c->ev_up); (void)hipEventDestroy(c->ev_compute);
  }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int lsdr_ctx_sync(lsdr_ctx *c) {
  LSDR_ARG(c);
  LSDR_HIP(hipStreamSynchronize(c->stream));
  return LSDR_OK;
}

void *lsdr_ctx_stream(lsdr_ctx *c) { return c ? (void *)c->stream : nullptr; }

}   // extern "C"

// ---- pinned bounce arena (lsdr_internal.h)
static int stage_reserve(lsdr_ctx *c, size_t bytes, char **slot) {
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (c->stage_used + need > c->stage_cap) {
    // outgrown: copies in flight keep using the old arena (freed at the next sync); the new one is twice as large
    size_t cap = c->stage_cap ? c->stage_cap * 2 : ((size_t)1 << 20);
    while (cap < need) cap *= 2;
    char *nb = nullptr;
    LSDR_HIP(hipHostMalloc((void **)&nb, cap, hipHostMallocDefault));
    if (c->stage_base) c->stage_retired.push_back(c->stage_base);
    c->stage_base = nb; c->stage_cap = cap; c->stage_used = 0;
  }
  *slot = c->stage_base + c->stage_used;
  c->stage_used += need;
  return LSDR_OK;
}
int lsdr_stage_h2d(lsdr_ctx *c, void *dst_dev, const void *src_host, size_t bytes) {
  LSDR_ARG(c);
  if (!bytes) return LSDR_OK;
  char *slot;
  { int rc = stage_reserve(c, bytes, &slot); if (rc) return rc; }
  memcpy(slot, src_host, bytes);
  LSDR_HIP(hipMemcpyAsync(dst_dev, slot, bytes, hipMemcpyHostToDevice, c->stream));
  return LSDR_OK;
}
int lsdr_stage_d2h(lsdr_ctx *c, void *dst_host, const void *src_dev, size_t bytes) {
  LSDR_ARG(c);
  if (!bytes) return LSDR_OK;
  char *slot;
  // A failure makes the caller return before its lsdr_stage_sync: the destinations queued so far (its locals) must not be
  // written by somebody else's next sync.
  { int rc = stage_reserve(c, bytes, &slot); if (rc) { c->stage_pending.clear(); return rc; } }
  const hipError_t e = hipMemcpyAsync(slot, src_dev, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e != hipSuccess) { c->stage_pending.clear(); LSDR_HIP(e); }
  c->stage_pending.push_back({dst_host, slot, bytes});
  return LSDR_OK;
}
int lsdr_stage_sync(lsdr_ctx *c) {
  LSDR_ARG(c);
  const hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess)
    for (const auto &p : c->stage_pending) memcpy(p.dst, p.src, p.bytes);
  c->stage_pending.clear();
  c->stage_used = 0;
  for (char *r : c->stage_retired) (void)hipHostFree(r);
  c->stage_retired.clear();
  LSDR_HIP(e);
  return LSDR_OK;
}

extern "C" {

int lsdr_malloc(lsdr_ctx *c, size_t bytes, void **p) {
  LSDR_ARG(c && p);
  LSDR_HIP(hipSetDevice(c->device));
  if (c->arena && bytes >= ((size_t)1 << 20) && lsdr_arena_malloc(c->arena, bytes, p) == LSDR_OK) return LSDR_OK;   // (a full arena: an ordinary allocation)
  LSDR_HIP(hipMalloc(p, bytes ? bytes : 1));
  return LSDR_OK;
}
int lsdr_free(lsdr_ctx *c, void *p) {
  LSDR_ARG(c);
  if (!p) return LSDR_OK;
  if (c->arena && lsdr_arena_owns(c->arena, p)) return lsdr_arena_release(c->arena, p);
  LSDR_HIP(hipStreamSynchronize(c->stream));
  LSDR_HIP(hipFree(p));
  return LSDR_OK;
}
int lsdr_malloc_host(size_t bytes, void **p) {
  LSDR_ARG(p);
  LSDR_HIP(hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
  return LSDR_OK;
}
int lsdr_free_host(void *p) {
  if (!p) return LSDR_OK;
  LSDR_HIP(hipHostFree(p));
  return LSDR_OK;
}
int lsdr_memcpy_h2d(lsdr_ctx *c, void *dst, const void *src, size_t bytes) {
  LSDR_ARG(c);
  if (!bytes) return LSDR_OK;
  LSDR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  return LSDR_OK;
}
int lsdr_memcpy_d2h(lsdr_ctx *c, void *dst, const void *src, size_t bytes) {
  LSDR_ARG(c);
  if (!bytes) return LSDR_OK;
  LSDR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  return LSDR_OK;
}
int lsdr_memset(lsdr_ctx *c, void *dst, int value, size_t bytes) {
  LSDR_ARG(c);
  if (!bytes) return LSDR_OK;
  LSDR_HIP(hipMemsetAsync(dst, value, bytes, c->stream));
  return LSDR_OK;
}

// pipebuf::pack() moves the unread tail to the front of the buffer
// (framework.h:153-159).  The ranges may overlap (dst < src), which a single
// device memcpy does not define; copy forward in chunks no larger than the gap.
int lsdr_memcpy_d2d(lsdr_ctx *c, void *dst, const void *src, size_t bytes) {
  LSDR_ARG(c);
  if (!bytes || dst == src) return LSDR_OK;
  char *d = (char *)dst;
  const char *s = (const char *)src;
  size_t gap = (d < s) ? (size_t)(s - d) : (size_t)(d - s);
  if (gap >= bytes) {
    LSDR_HIP(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, c->stream));
    return LSDR_OK;
  }
  // Overlapping ranges (pipebuf::pack() sliding a nearly full pipe by a small amount): two non-overlapping copies
  // through a bounce buffer.  Copying forward in gap-sized pieces instead is correct too but degenerates into
  // thousands of tiny copies when the shift is small (1.2 M copies / 3.9 s in a 157 M-sample run of leandvb_amd).
  if (c->bounce_cap < bytes) {
    LSDR_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(c->bounce);
    c->bounce = nullptr; c->bounce_cap = 0;
    LSDR_HIP(hipMalloc(&c->bounce, bytes));
    c->bounce_cap = bytes;
  }
  LSDR_HIP(hipMemcpyAsync(c->bounce, s, bytes, hipMemcpyDeviceToDevice, c->stream));
  LSDR_HIP(hipMemcpyAsync(d, c->bounce, bytes, hipMemcpyDeviceToDevice, c->stream));
  return LSDR_OK;
}

// ---- copy engine: host<->device transfers on side streams (north_star: "host<->device double-buffered hipMemcpyAsync on
// a side stream").  Uploads and downloads have a stream each (PCIe is full duplex); ordering against the compute stream is
// by events, never by blocking the host, except where a host reader needs the bytes (lsdr_copy_sync_d2h).
static int copy_engine(lsdr_ctx *c) {
  if (c->copy_ready) return LSDR_OK;
  LSDR_HIP(hipSetDevice(c->device));
  LSDR_HIP(hipStreamCreateWithFlags(&c->up, hipStreamNonBlocking));
  LSDR_HIP(hipStreamCreateWithFlags(&c->down, hipStreamNonBlocking));
  LSDR_HIP(hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming));
  LSDR_HIP(hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming));
  c->copy_ready = true;
  return LSDR_OK;
}
int lsdr_copy_h2d_async(lsdr_ctx *c, void *dst_dev, const void *src_host, size_t bytes) {
  LSDR_ARG(c);
  if (!bytes) return LSDR_OK;
  { int rc = copy_engine(c); if (rc) return rc; }
  LSDR_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->up));
  return LSDR_OK;
}
int lsdr_copy_d2h_async(lsdr_ctx *c, void *dst_host, const void *src_dev, size_t bytes) {
  LSDR_ARG(c);
  if (!bytes) return LSDR_OK;
  { int rc = copy_engine(c); if (rc) return rc; }
  LSDR_HIP(hipEventRecord(c->ev_compute, c->stream));          // after the kernels that produce the bytes
  LSDR_HIP(hipStreamWaitEvent(c->down, c->ev_compute, 0));
  LSDR_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->down));
  return LSDR_OK;
}
int lsdr_copy_fence(lsdr_ctx *c) {
  LSDR_ARG(c);
  if (!c->copy_ready) return LSDR_OK;
  LSDR_HIP(hipEventRecord(c->ev_up, c->up));
  LSDR_HIP(hipStreamWaitEvent(c->stream, c->ev_up, 0));
  return LSDR_OK;
}
int lsdr_copy_sync_d2h(lsdr_ctx *c) {
  LSDR_ARG(c);
  if (!c->copy_ready) return LSDR_OK;
  LSDR_HIP(hipStreamSynchronize(c->down));
  return LSDR_OK;
}
int lsdr_copy_sync_all(lsdr_ctx *c) {
  LSDR_ARG(c);
  if (c->copy_ready) { LSDR_HIP(hipStreamSynchronize(c->up)); }
  LSDR_HIP(hipStreamSynchronize(c->stream));
  if (c->copy_ready) { LSDR_HIP(hipStreamSynchronize(c->down)); }
  return LSDR_OK;
}

int lsdr_timer_start(lsdr_ctx *c) {
  LSDR_ARG(c);
  LSDR_HIP(hipEventRecord(c->ev0, c->stream));
  return LSDR_OK;
}
int lsdr_timer_stop_ms(lsdr_ctx *c, float *ms) {
  LSDR_ARG(c && ms);
  LSDR_HIP(hipEventRecord(c->ev1, c->stream));
  LSDR_HIP(hipEventSynchronize(c->ev1));
  LSDR_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return LSDR_OK;
}


int lsdr_event_create(lsdr_ctx *c, lsdr_event **ev) {
  LSDR_ARG(c && ev);
  lsdr_event *e = new lsdr_event();
  e->ctx = c;
  // Timing stays enabled (bench.py measures launches with these); the release at the event is agent-scope — every consumer is a
  // kernel on the same device — unless LSDR_EVENT_SYSTEM_FENCE asks for the default system-scope fence.
  {
    const char *sf = getenv("LSDR_EVENT_SYSTEM_FENCE");
    if (sf && atoi(sf)) LSDR_HIP(hipEventCreate(&e->ev));
    else LSDR_HIP(hipEventCreateWithFlags(&e->ev, hipEventDisableSystemFence));
  }
  *ev = e;
  return LSDR_OK;
}
void lsdr_event_destroy(lsdr_event *e) {
  if (!e) return;
  (void)hipEventDestroy(e->ev);
  delete e;
}
int lsdr_event_record(lsdr_event *e) {
  LSDR_ARG(e);
  LSDR_HIP(hipEventRecord(e->ev, e->ctx->stream));
  return LSDR_OK;
}
int lsdr_ctx_wait_event(lsdr_ctx *c, lsdr_event *e) {
  LSDR_ARG(c && e);
  LSDR_HIP(hipStreamWaitEvent(c->stream, e->ev, 0));
  return LSDR_OK;
}
int lsdr_event_elapsed_ms(lsdr_event *a, lsdr_event *b, float *ms) {
  LSDR_ARG(a && b && ms);
  LSDR_HIP(hipEventSynchronize(b->ev));
  LSDR_HIP(hipEventElapsedTime(ms, a->ev, b->ev));
  return LSDR_OK;
}

}  // extern "C"
