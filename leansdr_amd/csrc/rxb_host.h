// leansdr_amd/csrc/rxb_host.h — host side of the capture-batch receiver (rxb_device.h); included at the end of cstln_receiver.hip.
// Internal API (lsdr_internal.h) of lsdr_capture_batch (capture_batch.hip): create / launch / accessors / destroy.
#ifndef LSDR_RXB_HOST_H
#define LSDR_RXB_HOST_H

struct lsdr_rxb {
  lsdr_ctx *ctx;
  lsdr_rx *proto;                  // tables, loop constants, the constructed loop state, the relabel maps
  unsigned n;
  size_t max_samples;
  int anf;
  unsigned Lc, Wc, pre_block, pre_look;
  float nk;
  int notch_decimation;
  // capacities (for max_samples)
  unsigned max_tiles, max_det, max_pre, max_blocks, hwords;
  unsigned long long hpitch;
  size_t words_cap;                // 32-bit words of packed decisions per capture
  std::vector<rxb_cap> caps;       // host copy (buffer pointers are fixed at create; in / geometry per launch)
  rxb_cap *h_caps, *d_caps;        // pinned staging, device array
  rx_seam_result *d_res, *h_res; rx_state_dev *d_state_end, *d_state0; rx_ema_map *d_ema;   // h_res: pinned copy of d_res, filled behind every launch
  unsigned *d_iv_of_block, *d_det_block; float2 *d_om;
  std::vector<void *> owned;       // every per-capture device allocation
  size_t geom_samples;             // n_samples the detect-point tables on the device were built for
  unsigned n_det, n_pre, n_tiles; unsigned long long total_chunks;
  hipEvent_t tev0, tev1; bool timing; double time_ms; unsigned time_n; bool tev_pending;
  hipEvent_t ev_pre, ev_tiles;      // hand-overs between the tile stream and the auxiliary stream (lsdr_rxb_launch with aux)
};

static int rxb_alloc(lsdr_rxb *b, void **p, size_t bytes) {
  LSDR_HIP(hipMalloc(p, bytes ? bytes : 16));
  b->owned.push_back(*p);
  return LSDR_OK;
}

// geometry of a run over n_samples per capture
struct rxb_geom { unsigned long long chunks; unsigned n_tiles, n_det, n_pre, n_blocks; };
static rxb_geom rxb_geometry(const lsdr_rxb *b, size_t n_samples) {
  rxb_geom g;
  const size_t usable = b->anf ? n_samples / kDetN * kDetN : n_samples;      // auto_notch::run moves whole 4096-sample blocks (sdr.h:64-75)
  g.chunks = usable >= (size_t)(kChunk + 1) ? (usable - 1) / kChunk : 0;     // cstln_receiver::run: chunk_size + readahead readable (sdr.h:783)
  g.n_tiles = g.chunks ? 1u : 0u;
  if (g.chunks > b->Wc) g.n_tiles += (unsigned)((g.chunks - b->Wc + b->Lc - 1) / b->Lc);
  g.n_blocks = (unsigned)(usable / kDetN);
  g.n_det = 0;
  if (b->anf) {                                                              // `phase += fft.n; if (phase >= decimation) { phase -= decimation; detect(); }`
    long long phase = 0;
    for (unsigned blk = 0; blk < g.n_blocks; ++blk) { phase += kDetN; if (phase >= b->notch_decimation) { phase -= b->notch_decimation; ++g.n_det; } }
  }
  g.n_pre = b->anf ? (unsigned)(usable / b->pre_block) : 0u;
  return g;
}

int lsdr_rxb_create(lsdr_ctx *c, const lsdr_capture_batch_cfg *cfg, lsdr_rxb **out) {
  LSDR_ARG(c && cfg && out && cfg->n_captures >= 1 && cfg->n_captures <= 4096 && cfg->max_samples >= 4096);
  LSDR_ARG(cfg->anf == 0 || cfg->anf == 1);
  lsdr_rx_cfg rc;
  memset(&rc, 0, sizeof(rc));
  rc.sampler = LSDR_SAMP_LINEAR; rc.cstln = LSDR_QPSK; rc.fec = cfg->fec; rc.omega = cfg->omega; rc.freq = 0.f;
  rc.meas_decimation = 1048576; rc.pll_adjustment = 1.0f; rc.allow_drift = 0; rc.kest = 0.01f; rc.mode = LSDR_RX_TILED;
  rc.in_format = LSDR_IN_CU8; rc.out_format = LSDR_SYM_HARD2;
  if (!(cfg->omega >= 1.0f && cfg->omega <= 8.0f)) { lsdr_set_error("capture_batch: omega (samples per symbol) must be in [1, 8], got %g", (double)cfg->omega); return LSDR_E_UNSUPPORTED; }
  const unsigned L = cfg->tile_len ? cfg->tile_len : 4096u, W = cfg->tile_warmup ? cfg->tile_warmup : 512u;
  if (L % 2048u || W % kChunk || W < (unsigned)kChunk || (float)L / (cfg->omega + 0.1f) < 34.f) {
    lsdr_set_error("capture_batch: tile_len must be a multiple of 2048 samples and tile_warmup a multiple of 128 (got %u / %u)", L, W);
    return LSDR_E_ARG;
  }
  rc.tile_len = L; rc.tile_warmup = W;
  lsdr_rx *proto = nullptr;
  LSDR_TRY(lsdr_rx_create(c, &rc, &proto));
  lsdr_rxb *b = new lsdr_rxb();
  b->ctx = c; b->proto = proto; b->n = (unsigned)cfg->n_captures; b->max_samples = cfg->max_samples; b->anf = cfg->anf;
  b->Lc = L / kChunk; b->Wc = W / kChunk;
  b->pre_block = (L % 4096u) ? 2048u : 4096u;
  b->nk = cfg->notch_k > 0.f ? cfg->notch_k : 0.002f;                          // sdr.h:56
  b->notch_decimation = cfg->notch_decimation > 0 ? cfg->notch_decimation : 1024 * kDetN;
  {
    const double omk = (double)(1.0f - b->nk);
    unsigned q = 1;
    while (pow(omk, (double)q * b->pre_block) > 1e-7 && q < 64) ++q;
    b->pre_look = q;
  }
  *out = b;                                                                    // (from here on the caller destroys on error)
  LSDR_HIP(hipSetDevice(c->device));
  if (b->anf && (long long)b->Wc * kChunk > (long long)b->notch_decimation - kDetN) {
    lsdr_set_error("capture_batch: the first detect point must lie behind the first tile"); return LSDR_E_UNSUPPORTED;
  }
  const rxb_geom g = rxb_geometry(b, b->max_samples);
  b->max_tiles = g.n_tiles ? g.n_tiles : 1; b->max_det = g.n_det; b->max_pre = g.n_pre; b->max_blocks = g.n_blocks ? g.n_blocks : 1;
  const unsigned sym_per_chunk = (unsigned)(kChunk / (cfg->omega - 0.1f)) + 2;
  const unsigned stage_stride = (b->Wc > b->Lc ? b->Wc : b->Lc) * sym_per_chunk;
  b->hwords = stage_stride / 16 + 2;
  b->hpitch = ((unsigned long long)b->max_tiles + 63) & ~63ull;
  b->words_cap = (size_t)((unsigned long long)g.chunks * sym_per_chunk / 16 + b->max_tiles + 64);
  b->caps.assign(b->n, rxb_cap());
  LSDR_HIP(hipMalloc((void **)&b->d_caps, b->n * sizeof(rxb_cap)));
  LSDR_HIP(hipHostMalloc((void **)&b->h_caps, b->n * sizeof(rxb_cap), hipHostMallocDefault));
  LSDR_HIP(hipMalloc((void **)&b->d_res, b->n * sizeof(rx_seam_result)));
  LSDR_HIP(hipMemset(b->d_res, 0, b->n * sizeof(rx_seam_result)));
  LSDR_HIP(hipHostMalloc((void **)&b->h_res, b->n * sizeof(rx_seam_result), hipHostMallocDefault));
  memset(b->h_res, 0, b->n * sizeof(rx_seam_result));
  LSDR_HIP(hipMalloc((void **)&b->d_state_end, b->n * sizeof(rx_state_dev)));
  LSDR_HIP(hipMalloc((void **)&b->d_state0, sizeof(rx_state_dev)));
  LSDR_HIP(hipMemcpy(b->d_state0, &proto->st_initial, sizeof(rx_state_dev), hipMemcpyHostToDevice));
  LSDR_HIP(hipMalloc((void **)&b->d_ema, 2 * b->n * sizeof(rx_ema_map)));
  LSDR_HIP(hipMalloc((void **)&b->d_iv_of_block, (size_t)b->max_blocks * sizeof(unsigned)));
  LSDR_HIP(hipMalloc((void **)&b->d_det_block, (size_t)(b->max_det + 1) * sizeof(unsigned)));
  {
    std::vector<float2> om;
    notch_detect_twiddles(kDetN, true, om);
    LSDR_HIP(hipMalloc((void **)&b->d_om, kDetN * sizeof(float2)));
    LSDR_HIP(hipMemcpy(b->d_om, om.data(), kDetN * sizeof(float2), hipMemcpyHostToDevice));
  }
  for (unsigned i = 0; i < b->n; ++i) {
    rxb_cap &cp = b->caps[i];
    memset(&cp, 0, sizeof(cp));
    LSDR_TRY(rxb_alloc(b, (void **)&cp.hstage, (size_t)b->hpitch * b->hwords * sizeof(unsigned)));
    LSDR_TRY(rxb_alloc(b, (void **)&cp.hinfo, (size_t)b->max_tiles * sizeof(rx_tile_info_h)));
    LSDR_TRY(rxb_alloc(b, (void **)&cp.fix, (size_t)b->max_tiles * sizeof(rx_tile_fix)));
    LSDR_TRY(rxb_alloc(b, (void **)&cp.part, (size_t)((b->max_tiles + kSeamBlock - 1) / kSeamBlock) * sizeof(rx_seam_part)));
    LSDR_TRY(rxb_alloc(b, (void **)&cp.out_words, b->words_cap * sizeof(unsigned)));
    cp.res = b->d_res + i; cp.state_end = b->d_state_end + i; cp.ema_scratch = b->d_ema + 2 * i;
    cp.hpitch = b->hpitch;
    if (b->anf) {
      LSDR_TRY(rxb_alloc(b, (void **)&cp.iv, (size_t)(b->max_det + 1) * sizeof(rxb_iv)));
      LSDR_TRY(rxb_alloc(b, (void **)&cp.T, (size_t)(b->max_pre + 1) * sizeof(float2)));
      LSDR_TRY(rxb_alloc(b, (void **)&cp.cand, (size_t)(b->max_det + 1) * kDetMaxSlots * sizeof(int)));
      LSDR_TRY(rxb_alloc(b, (void **)&cp.halves, (size_t)(b->max_det + 1) * kDetN * sizeof(float2)));
    }
  }
  b->geom_samples = 0;
  LSDR_HIP(hipEventCreate(&b->tev0)); LSDR_HIP(hipEventCreate(&b->tev1));
  LSDR_HIP(hipEventCreateWithFlags(&b->ev_pre, hipEventDisableTiming)); LSDR_HIP(hipEventCreateWithFlags(&b->ev_tiles, hipEventDisableTiming));
  b->timing = false; b->time_ms = 0; b->time_n = 0; b->tev_pending = false;
  return LSDR_OK;
}

void lsdr_rxb_destroy(lsdr_rxb *b) {
  if (!b) return;
  (void)hipStreamSynchronize(b->ctx->stream);
  for (void *p : b->owned) (void)hipFree(p);
  (void)hipFree(b->d_caps); if (b->h_caps) (void)hipHostFree(b->h_caps);
  (void)hipFree(b->d_res); if (b->h_res) (void)hipHostFree(b->h_res); (void)hipFree(b->d_state_end); (void)hipFree(b->d_state0); (void)hipFree(b->d_ema);
  (void)hipFree(b->d_iv_of_block); (void)hipFree(b->d_det_block); (void)hipFree(b->d_om);
  if (b->tev0) (void)hipEventDestroy(b->tev0);
  if (b->tev1) (void)hipEventDestroy(b->tev1);
  if (b->ev_pre) (void)hipEventDestroy(b->ev_pre);
  if (b->ev_tiles) (void)hipEventDestroy(b->ev_tiles);
  lsdr_rx_destroy(b->proto);
  delete b;
}

static void rxb_fill_args(const lsdr_rxb *b, rxb_args &A) {
  A.caps = b->d_caps; A.iv_of_block = b->d_iv_of_block; A.det_block = b->d_det_block; A.om = b->d_om; A.state0 = b->d_state0;
  A.tile_chunks = b->Lc; A.warm_chunks = b->Wc; A.pre_block = b->pre_block; A.pre_look = b->pre_look;
  A.nk = b->nk; A.l2omk = (float)log2((double)(1.0f - b->nk));
  rx_fill_consts(b->proto, A.C, A.T);
}

// Queues the whole front end of a batch: detect chain, estimator pre-pass, tiles, seam pass, compaction.  aux == nullptr: everything on
// the context's stream.  aux: the TILES on the context's stream, everything else on `aux` (the caller's stream for the memory-bound
// kernels — on its own compute units, lsdr_capture_batch_cfg::aux_cus), handed over by events; what follows the compaction (the FEC tail)
// belongs on `aux` then.  The previous launch of this object must have completed (the argument records are single-buffered).
int lsdr_rxb_launch(lsdr_rxb *b, const void *const *iq, size_t n_samples, size_t *consumed, hipStream_t aux) {
  LSDR_ARG(b && iq && consumed);
  if (n_samples > b->max_samples) { lsdr_set_error("capture_batch: %zu samples per capture, created for %zu", n_samples, b->max_samples); return LSDR_E_ARG; }
  lsdr_ctx *c = b->ctx;
  lsdr_rx *r = b->proto;
  LSDR_HIP(hipSetDevice(c->device));
  if (b->tev_pending) LSDR_TRY(lsdr_rxb_tile_time(b, -1, nullptr, nullptr));   // (the previous launch has completed: collect its events)
  const rxb_geom g = rxb_geometry(b, n_samples);
  *consumed = (size_t)g.chunks * kChunk;
  b->n_det = g.n_det; b->n_pre = g.n_pre; b->n_tiles = g.n_tiles; b->total_chunks = g.chunks;
  if (b->anf && b->geom_samples != n_samples) {                               // detect points of this capture length (the same for every capture)
    std::vector<unsigned> ivb(g.n_blocks ? g.n_blocks : 1, 0u), det(g.n_det + 1, 0u);
    long long phase = 0;
    unsigned m = 0;
    for (unsigned blk = 0; blk < g.n_blocks; ++blk) {
      phase += kDetN;
      if (phase >= b->notch_decimation) { phase -= b->notch_decimation; det[m++] = blk; }
      ivb[blk] = m;
    }
    LSDR_HIP(hipStreamSynchronize(c->stream));
    if (aux) LSDR_HIP(hipStreamSynchronize(aux));
    LSDR_HIP(hipMemcpy(b->d_iv_of_block, ivb.data(), ivb.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    LSDR_HIP(hipMemcpy(b->d_det_block, det.data(), det.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    b->geom_samples = n_samples;
  }
  for (unsigned i = 0; i < b->n; ++i) {
    LSDR_ARG(iq[i] && ((unsigned long long)iq[i] & 1ull) == 0);
    rxb_cap &cp = b->caps[i];
    cp.in = static_cast<const unsigned char *>(iq[i]);
    cp.total_chunks = g.chunks; cp.n_tiles = g.n_tiles; cp.n_det = g.n_det;
    b->h_caps[i] = cp;
  }
  const hipStream_t sa = aux ? aux : c->stream, st = c->stream;      // auxiliary kernels / tiles
  LSDR_HIP(hipMemcpyAsync(b->d_caps, b->h_caps, b->n * sizeof(rxb_cap), hipMemcpyHostToDevice, sa));
  if (!g.chunks) {
    LSDR_HIP(hipMemsetAsync(b->d_res, 0, b->n * sizeof(rx_seam_result), sa));
    LSDR_HIP(hipMemcpyAsync(b->h_res, b->d_res, b->n * sizeof(rx_seam_result), hipMemcpyDeviceToHost, sa));
    return LSDR_OK;
  }
  rxb_args A;
  rxb_fill_args(b, A);
  const bool notch = b->anf && g.n_det > 0;
  if (notch) {
    if (((unsigned long long)iq[0] & 15ull) != 0) { lsdr_set_error("capture_batch: with the notch the captures must be 16-byte aligned"); return LSDR_E_ARG; }
    for (unsigned i = 1; i < b->n; ++i) LSDR_ARG(((unsigned long long)iq[i] & 15ull) == 0);
    hipLaunchKernelGGL(k_rxb_detect_fft, dim3(2 * g.n_det, b->n), dim3(256), 0, sa, A);
    hipLaunchKernelGGL(k_rxb_detect_peaks, dim3(g.n_det, b->n), dim3(256), 0, sa, A);
    hipLaunchKernelGGL(k_rxb_iv, dim3((b->n + 63) / 64), dim3(64), 0, sa, A, b->n);
    hipLaunchKernelGGL(k_rxb_notch_pre, dim3(g.n_pre, b->n), dim3(256), 0, sa, A);
    LSDR_HIP(hipGetLastError());
  }
  const unsigned blocks = 1 + (g.n_tiles - 1 + 63) / 64;
  if (aux) { LSDR_HIP(hipEventRecord(b->ev_pre, sa)); LSDR_HIP(hipStreamWaitEvent(st, b->ev_pre, 0)); }
  if (b->timing) { LSDR_HIP(hipEventRecord(b->tev0, st)); }
  if (notch) hipLaunchKernelGGL(k_rxb_tiles<true>, dim3(blocks, b->n), dim3(64), 0, st, A);
  else hipLaunchKernelGGL(k_rxb_tiles<false>, dim3(blocks, b->n), dim3(64), 0, st, A);
  if (b->timing) { LSDR_HIP(hipEventRecord(b->tev1, st)); b->tev_pending = true; }
  if (aux) { LSDR_HIP(hipEventRecord(b->ev_tiles, st)); LSDR_HIP(hipStreamWaitEvent(sa, b->ev_tiles, 0)); }
  const int R = r->tabs.nrotations;
  const float quad = 65536.0f / R;
  hipLaunchKernelGGL(k_rxb_seam, dim3((g.n_tiles + kSeamBlock - 1) / kSeamBlock, b->n), dim3(kSeamBlock), 0, sa, A, r->omega, R, quad,
                     (const uint8_t *)r->d_relabel);
  hipLaunchKernelGGL(k_rxb_compact, dim3((g.n_tiles * kRxbCompactLanes + 63) / 64, b->n), dim3(64), 0, sa, A, R, quad, (const uint8_t *)r->d_relabel);
  LSDR_HIP(hipGetLastError());
  LSDR_HIP(hipMemcpyAsync(b->h_res, b->d_res, b->n * sizeof(rx_seam_result), hipMemcpyDeviceToHost, sa));
  return LSDR_OK;
}

const uint32_t *lsdr_rxb_words(const lsdr_rxb *b, unsigned i) { return b && i < b->n ? b->caps[i].out_words : nullptr; }
size_t lsdr_rxb_words_cap(const lsdr_rxb *b) { return b ? b->words_cap : 0; }
// device array [n]: .total = packed decisions of capture i after the launch (struct rx_seam_result: 8-byte total first)
const void *lsdr_rxb_results_dev(const lsdr_rxb *b, size_t *stride) { if (stride) *stride = sizeof(rx_seam_result); return b ? b->d_res : nullptr; }
unsigned lsdr_rxb_tiles(const lsdr_rxb *b) { return b ? b->n_tiles : 0; }
unsigned lsdr_rxb_detects(const lsdr_rxb *b) { return b ? b->n_det : 0; }
// debugging / tests: the detected bins of capture i (synchronous)
int lsdr_rxb_bins(lsdr_rxb *b, unsigned i, int *bins, unsigned cap, unsigned *n) {
  LSDR_ARG(b && i < b->n && n);
  *n = b->anf ? b->n_det : 0;
  if (!*n || !bins) return LSDR_OK;
  LSDR_HIP(hipStreamSynchronize(b->ctx->stream));
  std::vector<int> cand((size_t)b->n_det * kDetMaxSlots);
  LSDR_HIP(hipMemcpy(cand.data(), b->caps[i].cand, cand.size() * sizeof(int), hipMemcpyDeviceToHost));
  for (unsigned q = 0; q < b->n_det && q < cap; ++q) bins[q] = cand[(size_t)q * kDetMaxSlots];
  return LSDR_OK;
}
// seam statistics of capture i's last launch; valid once the stream has passed the launch (the caller has waited for an event behind it)
int lsdr_rxb_seam_stats(lsdr_rxb *b, unsigned i, unsigned long long *total, unsigned *dup, unsigned *miss, unsigned *bad) {
  LSDR_ARG(b && i < b->n);
  const rx_seam_result sr = b->h_res[i];
  if (total) *total = sr.total;
  if (dup) *dup = sr.ndup;
  if (miss) *miss = sr.nmiss;
  if (bad) *bad = sr.nbad;
  return LSDR_OK;
}
// tests: the notched stream of capture i as the tiles of the LAST launch saw it (n cf32 items to device memory `out`); synchronous
int lsdr_rxb_notched(lsdr_rxb *b, unsigned i, lsdr_cf32 *out_dev, size_t n) {
  LSDR_ARG(b && i < b->n && out_dev);
  if (!b->anf || !b->geom_samples) { lsdr_set_error("capture_batch: no notch in this batch (or no run yet)"); return LSDR_E_ARG; }
  const size_t usable = b->geom_samples / kDetN * kDetN;
  if (n > usable) n = usable;
  rxb_args A;
  rxb_fill_args(b, A);
  const unsigned segs = (unsigned)((n + b->pre_block - 1) / b->pre_block);
  hipLaunchKernelGGL(k_rxb_notch_dump, dim3((segs + 63) / 64), dim3(64), 0, b->ctx->stream, A, i, (unsigned long long)n, reinterpret_cast<float2 *>(out_dev));
  LSDR_HIP(hipGetLastError());
  LSDR_HIP(hipStreamSynchronize(b->ctx->stream));
  return LSDR_OK;
}
int lsdr_rxb_tile_time(lsdr_rxb *b, int enable, float *avg_ms, unsigned *launches) {
  LSDR_ARG(b);
  if (b->tev_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(b->tev1) == hipSuccess && hipEventElapsedTime(&ms, b->tev0, b->tev1) == hipSuccess) { b->time_ms += ms; ++b->time_n; }
    b->tev_pending = false;
  }
  if (avg_ms) *avg_ms = b->time_n ? (float)(b->time_ms / b->time_n) : 0.f;
  if (launches) *launches = b->time_n;
  if (enable < 0) return LSDR_OK;      // (collect only)
  b->time_ms = 0; b->time_n = 0; b->timing = enable != 0;
  return LSDR_OK;
}

#endif  // LSDR_RXB_HOST_H
