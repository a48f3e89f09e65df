// leansdr_amd/csrc/tail_host.h — host side of the device-resident FEC tail (tail_device.h); included at the end of fec.hip.
// Internal API (lsdr_internal.h) of lsdr_capture_batch (capture_batch.hip).
#ifndef LSDR_TAIL_HOST_H
#define LSDR_TAIL_HOST_H

struct lsdr_tail {
  lsdr_ctx *ctx;
  unsigned n;
  size_t sym_cap, byte_cap, pk_cap;
  lsdr_deconv *dec;                 // the polynomials / alignment tables of deconvol_sync, built once by its own constructor
  lsdr_derandomizer *der;           // the PRBS pattern on the device
  tail_args A;
  std::vector<tail_cap> caps;       // host copy (pointers)
  tail_cap *d_caps;
  tail_result *h_res, *h_res_dev;   // pinned, [n]
  std::vector<void *> owned;
};

static int tail_alloc(lsdr_tail *t, void **p, size_t bytes) {
  LSDR_HIP(hipMalloc(p, bytes ? bytes : 16));
  t->owned.push_back(*p);
  return LSDR_OK;
}

int lsdr_tail_create(lsdr_ctx *c, unsigned n, size_t sym_cap, int rate, unsigned window, lsdr_tail **out) {
  LSDR_ARG(c && out && n >= 1 && sym_cap >= 1 && window >= 2048);      // (mpeg_sync's search needs 204·8 + 1 bytes in one call)
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_tail *t = new lsdr_tail();
  t->ctx = c; t->n = n; t->sym_cap = sym_cap;
  *out = t;                                                             // (from here on the caller destroys on error)
  LSDR_TRY(lsdr_deconv_create(c, rate, 0, &t->dec));
  LSDR_TRY(lsdr_derandomizer_create(c, &t->der));
  gf_tables *tab = rs_device_tables(c);
  if (!tab) { lsdr_set_error("capture_batch: cannot allocate GF tables"); return LSDR_E_NOMEM; }
  const deconv_host &H = t->dec->H;
  // bytes out of sym_cap symbols (rate pp/(pw/2) bits per symbol) with slack; packets of 204 bytes
  t->byte_cap = (size_t)((unsigned long long)sym_cap * (unsigned)H.pp / (unsigned)(H.pw / 2) / 8) + 65536;
  t->pk_cap = t->byte_cap / kRS + 64;
  memset(&t->A, 0, sizeof(t->A));
  for (int b = 0; b < 8; ++b) t->A.D.deconv[b] = b < H.pp ? H.deconv[b] : 0;
  t->A.D.pp = H.pp; t->A.D.pw = H.pw;
  for (int a = 0; a < 4; ++a) for (int s = 0; s < 4; ++s) t->A.luts[a][s] = t->dec->luts[a][s];
  memset(&t->A.ms0, 0, sizeof(t->A.ms0));
  t->A.ms0.scan_syncs = 8; t->A.ms0.want_syncs = 4; t->A.ms0.lock_timeout = 4; t->A.ms0.resync_period = 1;      // dvb.h:727-731
  t->A.gtab = tab;
  t->A.window = window;
  {
    uint8_t G[17];
    lsdr_rs_tables(nullptr, nullptr, G);                                // G[i] = coefficient of x^(16−i) (rs.h:93-105); G[0] = 1
    for (int m = 0; m < 16; ++m) t->A.rs_g[m] = G[1 + m];
  }
  t->A.pattern = t->der->d_pattern;
  t->caps.assign(n, tail_cap());
  LSDR_HIP(hipMalloc((void **)&t->d_caps, n * sizeof(tail_cap)));
  LSDR_HIP(hipHostMalloc((void **)&t->h_res, n * sizeof(tail_result), hipHostMallocDefault));
  LSDR_HIP(hipHostGetDevicePointer((void **)&t->h_res_dev, t->h_res, 0));
  memset(t->h_res, 0, n * sizeof(tail_result));
  for (unsigned i = 0; i < n; ++i) {
    tail_cap &tc = t->caps[i];
    memset(&tc, 0, sizeof(tc));
    LSDR_TRY(tail_alloc(t, (void **)&tc.bytes, t->byte_cap + 64));
    LSDR_TRY(tail_alloc(t, (void **)&tc.mpeg, t->byte_cap + 64));
    LSDR_TRY(tail_alloc(t, (void **)&tc.rs, (t->pk_cap + kRsChunk) * kRS + 64));        // (k_tail_rs stages 16-byte pieces: room behind the last packet)
    LSDR_TRY(tail_alloc(t, (void **)&tc.rts, t->pk_cap * kTS));
    LSDR_TRY(tail_alloc(t, (void **)&tc.ts, t->pk_cap * kTS));
    LSDR_TRY(tail_alloc(t, (void **)&tc.first, t->pk_cap + kRsChunk));
    LSDR_TRY(tail_alloc(t, (void **)&tc.pkt_pos, t->pk_cap * sizeof(int)));
    LSDR_TRY(tail_alloc(t, (void **)&tc.pkt_dst, t->pk_cap * sizeof(long long)));
    tc.byte_cap = t->byte_cap; tc.pk_cap = t->pk_cap;
    tc.res = t->h_res_dev + i;
  }
  t->A.caps = t->d_caps;
  return LSDR_OK;
}

void lsdr_tail_destroy(lsdr_tail *t) {
  if (!t) return;
  (void)hipStreamSynchronize(t->ctx->stream);
  for (void *p : t->owned) (void)hipFree(p);
  (void)hipFree(t->d_caps);
  if (t->h_res) (void)hipHostFree(t->h_res);
  lsdr_deconv_destroy(t->dec);
  lsdr_derandomizer_destroy(t->der);
  delete t;
}

// Inputs of capture i: its packed decisions and where their count will be (device memory, 8 bytes).  Uploads the records.
int lsdr_tail_bind(lsdr_tail *t, const uint32_t *const *words, const void *counts_dev, size_t count_stride) {
  LSDR_ARG(t && words && counts_dev);
  for (unsigned i = 0; i < t->n; ++i) {
    t->caps[i].words = words[i];
    t->caps[i].nsym = reinterpret_cast<const unsigned long long *>(static_cast<const char *>(counts_dev) + i * count_stride);
  }
  LSDR_HIP(hipMemcpy(t->d_caps, t->caps.data(), t->n * sizeof(tail_cap), hipMemcpyHostToDevice));
  return LSDR_OK;
}

// Queues the tail of every capture on the context's stream.  `before_ts`: an event the kernel that WRITES the TS buffers waits for (the
// download of the previous batch's TS), or null.
int lsdr_tail_launch(lsdr_tail *t, hipEvent_t before_ts) {
  LSDR_ARG(t);
  lsdr_ctx *c = t->ctx;
  LSDR_HIP(hipSetDevice(c->device));
  const dim3 one(1, t->n);
  // whole-chip kernels: enough workgroups for the largest capture, shared by the captures (a workgroup with nothing to do leaves at once)
  unsigned wide = (unsigned)c->num_cu * 8u / t->n;
  if (wide < 16) wide = 16;
  wide = (wide + 7) / 8 * 8;
  const dim3 grid(wide, t->n);
  hipLaunchKernelGGL(k_tail_acquire, one, dim3(256), 0, c->stream, t->A);
  hipLaunchKernelGGL(k_tail_deconv, grid, dim3(256), 0, c->stream, t->A);
  hipLaunchKernelGGL(k_tail_realign, grid, dim3(256), 0, c->stream, t->A);
  hipLaunchKernelGGL(k_tail_book, one, dim3(256), 0, c->stream, t->A);
  hipLaunchKernelGGL(k_tail_deint, grid, dim3(256), 0, c->stream, t->A);
  hipLaunchKernelGGL(k_tail_rs, grid, dim3(256), 0, c->stream, t->A);
  hipLaunchKernelGGL(k_tail_derand_scan, one, dim3(1024), 0, c->stream, t->A);
  LSDR_HIP(hipGetLastError());
  if (before_ts) LSDR_HIP(hipStreamWaitEvent(c->stream, before_ts, 0));
  hipLaunchKernelGGL(k_tail_derand_apply, grid, dim3(256), 0, c->stream, t->A);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

const lsdr_tail_result *lsdr_tail_results(const lsdr_tail *t) { return t ? reinterpret_cast<const lsdr_tail_result *>(t->h_res) : nullptr; }
const uint8_t *lsdr_tail_ts_dev(const lsdr_tail *t, unsigned i) { return t && i < t->n ? t->caps[i].ts : nullptr; }
size_t lsdr_tail_ts_cap(const lsdr_tail *t) { return t ? t->pk_cap * kTS : 0; }
// tests: the deconvolved bytes / the mpeg_sync output of capture i (device pointers; counts in the result record)
const uint8_t *lsdr_tail_bytes_dev(const lsdr_tail *t, unsigned i) { return t && i < t->n ? t->caps[i].bytes : nullptr; }
const uint8_t *lsdr_tail_mpeg_dev(const lsdr_tail *t, unsigned i) { return t && i < t->n ? t->caps[i].mpeg : nullptr; }

static_assert(sizeof(lsdr_tail_result) == sizeof(tail_result), "lsdr_internal.h mirrors tail_device.h's result record");

#endif  // LSDR_TAIL_HOST_H
