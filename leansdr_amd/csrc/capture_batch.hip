// leansdr_amd/csrc/capture_batch.hip — lsdr_capture_batch (include/lsdr_hip.h): B independent cu8 captures from their first sample to TS,
// one set of launches for all of them, counts on the device.  Host-side glue only: the front end is cstln_receiver.hip's lsdr_rxb
// (rxb_device.h / rxb_host.h), the FEC tail fec.hip's lsdr_tail (tail_device.h / tail_host.h).
//
// Replaces, per capture, one `leandvb --u8 -f Fs --sr Fm --cr R` process of the reference (leandvb.cc:157-600: its default graph).
#include "lsdr_internal.h"

struct lsdr_capture_batch {
  lsdr_ctx *ctx;                   // the context the front end's tiles run on: the caller's, or (aux_cus) an own one on the tile partition
  lsdr_ctx *ctx_aux;               // aux_cus: the own context of the auxiliary partition (null: everything on ctx)
  lsdr_ctx *own_main;              // aux_cus: ctx is ours
  lsdr_capture_batch_cfg cfg;
  lsdr_rxb *rx;
  lsdr_tail *tail;
  hipEvent_t ev_done, ev_dl;
  hipStream_t dl;                  // TS downloads
  bool in_flight, dl_pending, waited;
  size_t consumed;
  std::vector<unsigned long long> n_ts;
};

extern "C" {

void lsdr_capture_batch_destroy(lsdr_capture_batch *b) {
  if (!b) return;
  if (b->dl) { (void)hipStreamSynchronize(b->dl); }
  if (b->ctx) (void)hipStreamSynchronize(b->ctx->stream);
  if (b->ctx_aux) (void)hipStreamSynchronize(b->ctx_aux->stream);
  lsdr_tail_destroy(b->tail);
  lsdr_rxb_destroy(b->rx);
  if (b->ctx_aux) lsdr_ctx_destroy(b->ctx_aux);
  if (b->own_main) lsdr_ctx_destroy(b->own_main);
  if (b->ev_done) (void)hipEventDestroy(b->ev_done);
  if (b->ev_dl) (void)hipEventDestroy(b->ev_dl);
  if (b->dl) (void)hipStreamDestroy(b->dl);
  delete b;
}

// compute-unit masks of the two partitions: mask bit i = CU i % 32 of XCD i / 32 (8 XCDs of 32 CUs); the auxiliary partition takes the
// first aux/8 CUs of every XCD, the tiles the rest
static int capture_batch_partition(lsdr_capture_batch *b, lsdr_ctx *caller) {
  const unsigned aux = b->cfg.aux_cus, ncu = (unsigned)caller->num_cu;
  if (aux % 8 || aux >= ncu || ncu % 8 || ncu > 1024) { lsdr_set_error("capture_batch: aux_cus must be a multiple of 8 below the device's %u CUs (got %u)", ncu, aux); return LSDR_E_ARG; }
  const unsigned per = ncu / 8, words = (ncu + 31) / 32;
  std::vector<uint32_t> m_aux(words, 0u), m_main(words, 0u);
  for (unsigned i = 0; i < ncu; ++i) ((i % per) < aux / 8 ? m_aux : m_main)[i / 32] |= 1u << (i % 32);
  LSDR_TRY(lsdr_ctx_create_masked(caller->device, m_main.data(), words, &b->own_main));
  LSDR_TRY(lsdr_ctx_create_masked(caller->device, m_aux.data(), words, &b->ctx_aux));
  b->ctx = b->own_main;
  return LSDR_OK;
}

static int capture_batch_build(lsdr_capture_batch *b) {
  LSDR_HIP(hipSetDevice(b->ctx->device));
  if (b->cfg.aux_cus) LSDR_TRY(capture_batch_partition(b, b->ctx));
  lsdr_ctx *c = b->ctx;
  LSDR_TRY(lsdr_rxb_create(c, &b->cfg, &b->rx));
  const size_t sym_cap = lsdr_rxb_words_cap(b->rx) * 16;
  LSDR_TRY(lsdr_tail_create(b->ctx_aux ? b->ctx_aux : c, (unsigned)b->cfg.n_captures, sym_cap, b->cfg.fec, b->cfg.unlocked_window ? b->cfg.unlocked_window : 8192u, &b->tail));
  std::vector<const uint32_t *> words(b->cfg.n_captures);
  for (int i = 0; i < b->cfg.n_captures; ++i) words[i] = lsdr_rxb_words(b->rx, (unsigned)i);
  size_t stride = 0;
  const void *counts = lsdr_rxb_results_dev(b->rx, &stride);
  LSDR_TRY(lsdr_tail_bind(b->tail, words.data(), counts, stride));
  LSDR_HIP(hipEventCreateWithFlags(&b->ev_done, hipEventDisableTiming));
  LSDR_HIP(hipEventCreateWithFlags(&b->ev_dl, hipEventDisableTiming));
  LSDR_HIP(hipStreamCreateWithFlags(&b->dl, hipStreamNonBlocking));
  b->n_ts.assign(b->cfg.n_captures, 0);
  return LSDR_OK;
}

int lsdr_capture_batch_create(lsdr_ctx *c, const lsdr_capture_batch_cfg *cfg, lsdr_capture_batch **out) {
  LSDR_ARG(c && cfg && out);
  lsdr_capture_batch *b = new lsdr_capture_batch();
  b->ctx = c; b->cfg = *cfg;
  const int rc = capture_batch_build(b);
  if (rc) { lsdr_capture_batch_destroy(b); return rc; }
  *out = b;
  return LSDR_OK;
}

int lsdr_capture_batch_run_async(lsdr_capture_batch *b, const lsdr_cu8 *const *iq_dev, size_t n_samples) {
  LSDR_ARG(b && iq_dev);
  if (b->in_flight) { lsdr_set_error("capture_batch: a batch is in flight (lsdr_capture_batch_wait first)"); return LSDR_E_ARG; }
  LSDR_TRY(lsdr_rxb_launch(b->rx, reinterpret_cast<const void *const *>(iq_dev), n_samples, &b->consumed, b->ctx_aux ? b->ctx_aux->stream : nullptr));
  LSDR_TRY(lsdr_tail_launch(b->tail, b->dl_pending ? b->ev_dl : nullptr));
  LSDR_HIP(hipEventRecord(b->ev_done, (b->ctx_aux ? b->ctx_aux : b->ctx)->stream));
  b->in_flight = true; b->waited = false;
  return LSDR_OK;
}

int lsdr_capture_batch_wait(lsdr_capture_batch *b, lsdr_capture_result *results) {
  LSDR_ARG(b);
  if (!b->in_flight) { lsdr_set_error("capture_batch: no batch in flight"); return LSDR_E_ARG; }
  LSDR_HIP(hipEventSynchronize(b->ev_done));
  b->in_flight = false; b->waited = true;
  const lsdr_tail_result *tr = lsdr_tail_results(b->tail);
  for (int i = 0; i < b->cfg.n_captures; ++i) {
    b->n_ts[i] = tr[i].n_ts;
    if (!results) continue;
    lsdr_capture_result &r = results[i];
    memset(&r, 0, sizeof(r));
    r.ts_packets = tr[i].n_ts; r.rs_packets = tr[i].n_rs; r.rs_bit_errors = tr[i].rs_bit_errors; r.symbols = tr[i].symbols;
    r.samples = b->consumed; r.bytes_deconv = tr[i].bytes_deconv; r.bytes_mpeg = tr[i].bytes_mpeg; r.first_lock_byte = tr[i].first_lock_byte;
    r.next_sync_calls = tr[i].next_sync_calls; r.locked = tr[i].locked_at_end; r.alignment = tr[i].alignment; r.bitphase = tr[i].bitphase;
    r.tiles = lsdr_rxb_tiles(b->rx);
    unsigned long long tot = 0; unsigned d = 0, m = 0, bad = 0;
    LSDR_TRY(lsdr_rxb_seam_stats(b->rx, (unsigned)i, &tot, &d, &m, &bad));
    r.seam_dup = d; r.seam_miss = m; r.seam_bad = bad;
  }
  return LSDR_OK;
}

int lsdr_capture_batch_ts_download_async(lsdr_capture_batch *b, uint8_t *const *ts_host, size_t cap_bytes) {
  LSDR_ARG(b && ts_host);
  if (!b->waited) { lsdr_set_error("capture_batch: TS download before lsdr_capture_batch_wait"); return LSDR_E_ARG; }
  for (int i = 0; i < b->cfg.n_captures; ++i) {
    const size_t bytes = (size_t)b->n_ts[i] * 188;
    if (bytes > cap_bytes) { lsdr_set_error("capture_batch: capture %d has %zu TS bytes, the host buffer %zu", i, bytes, cap_bytes); return LSDR_E_ARG; }
    if (bytes) LSDR_HIP(hipMemcpyAsync(ts_host[i], lsdr_tail_ts_dev(b->tail, (unsigned)i), bytes, hipMemcpyDeviceToHost, b->dl));
  }
  LSDR_HIP(hipEventRecord(b->ev_dl, b->dl));
  b->dl_pending = true;
  return LSDR_OK;
}

int lsdr_capture_batch_ts_wait(lsdr_capture_batch *b) {
  LSDR_ARG(b);
  if (b->dl_pending) LSDR_HIP(hipEventSynchronize(b->ev_dl));
  b->dl_pending = false;
  return LSDR_OK;
}

const uint8_t *lsdr_capture_batch_ts_dev(const lsdr_capture_batch *b, int i) { return b && i >= 0 ? lsdr_tail_ts_dev(b->tail, (unsigned)i) : nullptr; }
const uint32_t *lsdr_capture_batch_words_dev(const lsdr_capture_batch *b, int i) { return b && i >= 0 ? lsdr_rxb_words(b->rx, (unsigned)i) : nullptr; }
const uint8_t *lsdr_capture_batch_bytes_dev(const lsdr_capture_batch *b, int i) { return b && i >= 0 ? lsdr_tail_bytes_dev(b->tail, (unsigned)i) : nullptr; }
const uint8_t *lsdr_capture_batch_mpeg_dev(const lsdr_capture_batch *b, int i) { return b && i >= 0 ? lsdr_tail_mpeg_dev(b->tail, (unsigned)i) : nullptr; }

int lsdr_capture_batch_bins(lsdr_capture_batch *b, int i, int *bins, unsigned cap, unsigned *n) {
  LSDR_ARG(b && i >= 0 && i < b->cfg.n_captures);
  return lsdr_rxb_bins(b->rx, (unsigned)i, bins, cap, n);
}
int lsdr_capture_batch_notched(lsdr_capture_batch *b, int i, lsdr_cf32 *out_dev, size_t n) {
  LSDR_ARG(b && i >= 0 && i < b->cfg.n_captures);
  if (b->in_flight) { lsdr_set_error("capture_batch: a batch is in flight (lsdr_capture_batch_wait first)"); return LSDR_E_ARG; }
  return lsdr_rxb_notched(b->rx, (unsigned)i, out_dev, n);
}
int lsdr_capture_batch_tile_time(lsdr_capture_batch *b, int enable, float *avg_ms, unsigned *launches) {
  LSDR_ARG(b);
  return lsdr_rxb_tile_time(b->rx, enable, avg_ms, launches);
}

}  // extern "C"
