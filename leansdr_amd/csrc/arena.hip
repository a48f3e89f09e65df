// leansdr_amd/csrc/arena.hip — lsdr_arena: PLACED stream buffers (include/lsdr_hip.h).
//
// WHERE a resident buffer lands in HBM decides how fast a streaming kernel reads it: one process, six 2 GiB buffers allocated one after
// the other — the same fir_filter launch takes 0.37 ms over some and 0.42 ms over others, reproducibly per buffer, whatever the
// allocation's flags (profiles/r05_bench/placement_probe*.txt).  It goes with the physical backing: the windows of ONE large
// allocation are of one kind far more often than separate allocations are, and the larger the allocation the larger the share of fast
// windows (96 GiB: all fast; 16 GiB: either — tools/placement_map.py).  Rounds 4-5 chose buffers in bench.py; the graph's own pipes
// (pipebuf storage, framework.h:141-143 → lsdr_malloc) got whatever came.  This is that choice as part of the library:
//   * an arena = ONE hipMalloc, handed out in 2 MiB-aligned windows;
//   * lsdr_arena_place carves the n fastest of up to max_windows free candidate windows — fastest under a PROBE the caller supplies (it
//     queues the launch whose speed matters on the context's stream; the library times it with events: 3 warm-up calls, 6 timed) or,
//     with no probe, under a built-in streaming read of the window.  The search stops early once a candidate is clearly of the fast
//     kind (at least five tried, the best 8 % under their median: the first measurements of a process run slow whatever the buffer, so
//     the slowest one is no yardstick);
//   * lsdr_ctx_set_arena makes lsdr_malloc serve every request of 1 MiB or more from the arena (built-in probe), so that a graph built on
//     the host framework gets placed pipes without knowing about it (LSDR_ARENA_GIB in the environment of a host application).
#include "lsdr_internal.h"

#include <algorithm>
#include <vector>

struct lsdr_arena {
  lsdr_ctx *ctx;
  char *base;
  size_t bytes;
  struct span { size_t off, len; };
  std::vector<span> used;                // sorted by offset, disjoint
  std::vector<float> log;                // probe time of every candidate of the last lsdr_arena_place
  // what the arena has LEARNT about itself: every probed window with its time relative to the median of its own lsdr_arena_place call
  // (< 1: faster than typical).  Later calls try the free positions inside known-fast stretches first, unknown ones next, known-slow
  // ones last: the slow kind comes in stretches of GiBs (one run: input windows 31–39 of 40 slow, and all 64 small windows of the tail
  // beside them slow too — while 30 small windows fit into ONE fast 2 GiB window that was not chosen).
  struct rec { size_t off, len; float rel; };
  std::vector<rec> map;
  unsigned *d_sink;                      // the built-in probe's result words
  hipEvent_t e0, e1;
};

namespace {
constexpr size_t kGran = (size_t)2 << 20;
inline size_t round_up(size_t v, size_t g) { return (v + g - 1) / g * g; }

// the built-in probe: every lane streams 16-byte loads over the window (non-temporal, like the filters' sample loads), one word per workgroup out
typedef unsigned arena_v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_arena_read(const arena_v4u *p, size_t n16, unsigned *sink) {
  arena_v4u acc = {0u, 0u, 0u, 0u};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) acc ^= __builtin_nontemporal_load(p + i);
  const unsigned r = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (r == 0x9e3779b9u) sink[blockIdx.x] = r;      // (never elided, practically never taken)
}

bool span_free(const lsdr_arena *a, size_t off, size_t len) {
  if (off + len > a->bytes) return false;
  for (const auto &s : a->used)
    if (off < s.off + s.len && s.off < off + len) return false;
  return true;
}
void span_take(lsdr_arena *a, size_t off, size_t len) {
  a->used.push_back({off, len});
  std::sort(a->used.begin(), a->used.end(), [](const lsdr_arena::span &x, const lsdr_arena::span &y) { return x.off < y.off; });
}
}  // namespace

extern "C" {

int lsdr_arena_create(lsdr_ctx *c, size_t bytes, lsdr_arena **out) {
  LSDR_ARG(c && out && bytes >= kGran);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_arena *a = new lsdr_arena();
  a->ctx = c; a->bytes = bytes / kGran * kGran; a->base = nullptr; a->d_sink = nullptr; a->e0 = a->e1 = nullptr;
  hipError_t e = hipMalloc((void **)&a->base, a->bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); delete a; lsdr_set_error("lsdr_arena_create: no %zu bytes of device memory in one piece", bytes); return LSDR_E_NOMEM; }
  if (hipMalloc((void **)&a->d_sink, 65536 * sizeof(unsigned)) != hipSuccess || hipEventCreate(&a->e0) != hipSuccess || hipEventCreate(&a->e1) != hipSuccess) {
    (void)hipGetLastError();
    lsdr_arena_destroy(a);
    return LSDR_E_HIP;
  }
  *out = a;
  return LSDR_OK;
}

void lsdr_arena_destroy(lsdr_arena *a) {
  if (!a) return;
  if (a->ctx) (void)hipStreamSynchronize(a->ctx->stream);
  if (a->ctx && a->ctx->arena == a) a->ctx->arena = nullptr;
  if (a->e0) (void)hipEventDestroy(a->e0);
  if (a->e1) (void)hipEventDestroy(a->e1);
  (void)hipFree(a->d_sink);
  (void)hipFree(a->base);
  delete a;
}

size_t lsdr_arena_bytes(const lsdr_arena *a) { return a ? a->bytes : 0; }
int lsdr_arena_owns(const lsdr_arena *a, const void *p) { return a && (const char *)p >= a->base && (const char *)p < a->base + a->bytes; }

int lsdr_arena_place(lsdr_arena *a, size_t bytes, unsigned n_best, unsigned max_windows, int from_tail, const void *fill_from, lsdr_probe_fn probe,
                     void *user, void **out, float *ms_out) {
  LSDR_ARG(a && bytes >= 1 && n_best >= 1 && n_best <= 16 && out);
  lsdr_ctx *c = a->ctx;
  LSDR_HIP(hipSetDevice(c->device));
  const size_t step = round_up(bytes, kGran);
  if (max_windows < n_best) max_windows = n_best;
  // candidates: the free windows on the grid of `step` from the arena's start (or, from_tail, from its end downwards)
  std::vector<size_t> cand;
  std::vector<float> t;
  a->log.clear();
  const size_t n_grid = a->bytes / step;
  auto stop = [&]() {
    if (t.size() < 5 || t.size() < n_best) return false;
    std::vector<float> s(t);
    std::sort(s.begin(), s.end());
    return s[n_best - 1] < 0.92f * s[s.size() / 2];          // the n-th best is clearly of the fast kind
  };
  // free grid positions, ordered: inside stretches known to be fast first (smallest relative time), never-probed ones as typical (1.0), known-slow
  // ones last; ties in address order (from the end with from_tail)
  struct pos { size_t off; float key; };
  std::vector<pos> order_in;
  for (size_t g = 0; g < n_grid; ++g) {
    const size_t off = from_tail ? a->bytes - (g + 1) * step : g * step;
    if (!span_free(a, off, step)) continue;
    float key = -1.f;
    for (const auto &r : a->map)
      if (off < r.off + r.len && r.off < off + step) key = key < r.rel ? r.rel : key;      // (the slowest stretch it touches)
    order_in.push_back({off, key < 0.f ? 1.0f : key});
  }
  std::stable_sort(order_in.begin(), order_in.end(), [](const pos &x, const pos &y) { return x.key < y.key; });
  for (size_t gi = 0; gi < order_in.size() && cand.size() < max_windows && !stop(); ++gi) {
    const size_t off = order_in[gi].off;
    char *w = a->base + off;
    if (fill_from) LSDR_HIP(hipMemcpyAsync(w, fill_from, bytes, hipMemcpyDeviceToDevice, c->stream));
    auto once = [&]() -> int {
      if (probe) return probe(user, w);
      hipLaunchKernelGGL(k_arena_read, dim3((unsigned)c->num_cu * 8), dim3(256), 0, c->stream, (const arena_v4u *)w, bytes / 16, a->d_sink);
      return hipGetLastError() == hipSuccess ? LSDR_OK : LSDR_E_HIP;
    };
    for (int i = 0; i < 3; ++i) LSDR_TRY(once());
    LSDR_HIP(hipEventRecord(a->e0, c->stream));
    for (int i = 0; i < 6; ++i) LSDR_TRY(once());
    LSDR_HIP(hipEventRecord(a->e1, c->stream));
    LSDR_HIP(hipEventSynchronize(a->e1));
    float ms = 0.f;
    LSDR_HIP(hipEventElapsedTime(&ms, a->e0, a->e1));
    cand.push_back(off); t.push_back(ms / 6); a->log.push_back(ms / 6);
  }
  if (cand.size() < n_best) { lsdr_set_error("lsdr_arena_place: %zu free window(s) of %zu bytes, %u asked for", cand.size(), step, n_best); return LSDR_E_NOMEM; }
  {
    std::vector<float> sm(t);
    std::sort(sm.begin(), sm.end());
    const float med = sm[sm.size() / 2] > 0.f ? sm[sm.size() / 2] : 1.f;
    for (size_t i = 0; i < cand.size(); ++i) a->map.push_back({cand[i], step, t[i] / med});
  }
  std::vector<size_t> order(cand.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return t[x] < t[y]; });
  for (unsigned k = 0; k < n_best; ++k) {
    span_take(a, cand[order[k]], step);
    out[k] = a->base + cand[order[k]];
    if (ms_out) ms_out[k] = t[order[k]];
  }
  return LSDR_OK;
}

// The probe's time over ANY device pointer (a buffer the caller already has: is the incumbent faster than the arena's best window?), measured
// like a candidate's: 3 untimed calls, 6 timed ones.
int lsdr_arena_time(lsdr_arena *a, void *ptr, lsdr_probe_fn probe, void *user, float *ms_out) {
  LSDR_ARG(a && ptr && probe && ms_out);
  lsdr_ctx *c = a->ctx;
  LSDR_HIP(hipSetDevice(c->device));
  for (int i = 0; i < 3; ++i) LSDR_TRY(probe(user, ptr));
  LSDR_HIP(hipEventRecord(a->e0, c->stream));
  for (int i = 0; i < 6; ++i) LSDR_TRY(probe(user, ptr));
  LSDR_HIP(hipEventRecord(a->e1, c->stream));
  LSDR_HIP(hipEventSynchronize(a->e1));
  float ms = 0.f;
  LSDR_HIP(hipEventElapsedTime(&ms, a->e0, a->e1));
  *ms_out = ms / 6;
  return LSDR_OK;
}

int lsdr_arena_release(lsdr_arena *a, void *window) {
  LSDR_ARG(a);
  if (!window) return LSDR_OK;
  LSDR_HIP(hipStreamSynchronize(a->ctx->stream));
  const size_t off = (size_t)((char *)window - a->base);
  for (size_t i = 0; i < a->used.size(); ++i)
    if (a->used[i].off == off) { a->used.erase(a->used.begin() + (long)i); return LSDR_OK; }
  lsdr_set_error("lsdr_arena_release: %p is not a window of this arena", window);
  return LSDR_E_ARG;
}

int lsdr_arena_probe_log(const lsdr_arena *a, float *ms, unsigned cap, unsigned *n) {
  LSDR_ARG(a && n);
  *n = (unsigned)a->log.size();
  for (unsigned i = 0; i < cap && i < *n; ++i) ms[i] = a->log[i];
  return LSDR_OK;
}

int lsdr_ctx_set_arena(lsdr_ctx *c, lsdr_arena *a) {
  LSDR_ARG(c && (!a || a->ctx == c));
  c->arena = a;
  return LSDR_OK;
}

}  // extern "C"

// lsdr_malloc / lsdr_free's side (ctx.hip): a request of 1 MiB or more goes to the context's arena while it has room
int lsdr_arena_malloc(lsdr_arena *a, size_t bytes, void **p) {
  void *w = nullptr;
  const int rc = lsdr_arena_place(a, bytes, 1, 12, 0, nullptr, nullptr, nullptr, &w, nullptr);
  if (rc) return rc;
  *p = w;
  return LSDR_OK;
}
