// leansdr_amd/csrc/tx.hip — the transmit chain of leandvbtx (leandvbtx.cc:79-175) on gfx950: the generator side of the
// hot path, so that synthetic DVB-S signals can be produced where they are consumed.
//
//   randomizer          dvb.h:1063-1102   XOR with the 8-packet PRBS pattern
//   rs_encoder          dvb.h:957-980, rs.h:141-167   16 parity bytes per 188-byte packet (polynomial division by G)
//   interleaver         dvb.h:899-921     Forney I=12: out[p][i] = in[p + 11 − i%12][i]
//   dvb_convol          dvb.h:567-604, convolutional.h:226-270   K=7 mother code + puncturing by polynomial shifts
//   cstln_transmitter   sdr.h:1196-1222   symbol → constellation point
//   fir_resampler       dsp.h:290-364     polyphase interpolator, complex·complex taps in the reference's order
//   simple_agc          sdr.h:238-274     per-128-sample power estimate, EMA, gain
//
// All integer/byte blocks are data-parallel given a few carried bytes; the AGC's per-chunk sums and its EMA are
// sequential float recurrences and are evaluated in exactly the reference's order (a lane per chunk, then one lane
// over the chunks).  Bit-exact against the oracle, which is pinned to the reference blocks and to `leandvbtx` itself.
#include <mutex>
#include "lsdr_internal.h"

namespace {

constexpr int kTS = 188, kRS = 204;

__global__ __launch_bounds__(256) void k_randomize(const unsigned char *in, unsigned long long nbytes, const unsigned char *pattern,
                                                   unsigned pos0, unsigned char *out) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nbytes; i += stride)
    out[i] = in[i] ^ pattern[(pos0 + i) % 1504u];
}

struct gf_tab { unsigned char exp[512], log[256], G[17]; };

// rs_engine::encode (rs.h:141-167): one thread per packet, tables in LDS.
__global__ __launch_bounds__(64) void k_rs_encode(const unsigned char *in, unsigned long long npackets, const gf_tab *tab,
                                                  unsigned char *out) {
  __shared__ gf_tab g;
  for (unsigned i = threadIdx.x; i < sizeof(gf_tab); i += 64) ((unsigned char *)&g)[i] = ((const unsigned char *)tab)[i];
  __syncthreads();
  const unsigned long long p = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
  if (p >= npackets) return;
  const unsigned char *pin = in + p * kTS;
  unsigned char *po = out + p * kRS;
  unsigned char r[17];   // sliding remainder: r[0] pairs with message byte d
  for (int i = 0; i < 17; ++i) r[i] = 0;
  auto gmul = [&](unsigned char x, unsigned char y) -> unsigned char { return (!x || !y) ? 0 : g.exp[g.log[x] + g.log[y]]; };
  const unsigned char g0 = g.G[0];
  for (int d = 0; d < kTS; ++d) {
    const unsigned char m = pin[d];
    po[d] = m;
    const unsigned char top = m ^ r[0];            // p[d] of the reference after the earlier XORs
    for (int i = 0; i < 16; ++i) r[i] = r[i + 1];
    r[16] = 0;
    if (top) {
      const unsigned char k = g.exp[g.log[top] + 255 - g.log[g0]];   // gdiv(p[d], G[0])
      for (int i = 1; i <= 16; ++i) r[i - 1] ^= gmul(k, g.G[i]);
    }
  }
  for (int i = 0; i < 16; ++i) po[kTS + i] = r[i];
}

__global__ __launch_bounds__(256) void k_interleave(const unsigned char *in, unsigned long long nout_packets, unsigned char *out) {
  const unsigned long long total = nout_packets * kRS, stride = (unsigned long long)gridDim.x * 256;
  for (unsigned long long j = (unsigned long long)blockIdx.x * 256 + threadIdx.x; j < total; j += stride) {
    const unsigned long long p = j / kRS;
    const unsigned i = (unsigned)(j % kRS);
    out[j] = in[(p + 11 - i % 12) * kRS + i];
  }
}

struct convol_args {
  const unsigned char *in; unsigned long long nbytes;
  unsigned char *out; unsigned long long nsym;
  int bits_in, bits_out, bps;
  unsigned short polys[8];
  unsigned short hist0;      // convol_multipoly::hist at the start of the call
};
// hist after input bit t (MSB-first bit stream): bit 15−k = input bit t−k (convolutional.h:244); older bits from hist0
__device__ __forceinline__ unsigned conv_hist(const convol_args &a, long long t) {
  unsigned h = 0;
  for (int k = 0; k < 16; ++k) {
    const long long u = t - k;
    unsigned bit;
    if (u >= 0) bit = (a.in[u >> 3] >> (7 - (int)(u & 7))) & 1u;
    else bit = (a.hist0 >> (15 - (int)(-u - 1))) & 1u;     // hist0 bit 15 = input bit −1, bit 14 = −2, …
    h |= bit << (15 - k);
  }
  return h;
}
__global__ __launch_bounds__(256) void k_convol(convol_args a) {
  const unsigned long long s = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (s >= a.nsym) return;
  unsigned v = 0;
  for (int b = 0; b < a.bps; ++b) {
    const unsigned long long o = s * (unsigned)a.bps + (unsigned)b;       // index into the coded bit stream
    const unsigned long long g = o / (unsigned)a.bits_out;
    const int p = (int)(o % (unsigned)a.bits_out);
    const long long t = (long long)((g + 1) * (unsigned)a.bits_in) - 1;   // last input bit of group g
    v = (v << 1) | (unsigned)(__popc(conv_hist(a, t) & a.polys[p]) & 1);
  }
  a.out[s] = (unsigned char)v;
}

__global__ __launch_bounds__(256) void k_cstln_map(const unsigned char *sym, unsigned long long n, const float2 *points, float2 *out) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = points[sym[i]];
}

// fir_resampler::run (dsp.h:318-332): out[m·interp + i] = Σ_k sc[i + k·interp]·in[latency + m − k], k ascending, x = x + c·p
__global__ __launch_bounds__(256) void k_fir_resample(const float2 *in, const float2 *sc, unsigned ncoeffs, unsigned interp, unsigned latency,
                                                      unsigned long long nout, float2 *out) {
  const unsigned long long o = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= nout) return;
  const unsigned long long m = o / interp;
  const unsigned i = (unsigned)(o % interp);
  const float2 *pi = in + latency + m;
  float xr = 0.f, xi = 0.f;
  for (unsigned c = i; c < ncoeffs; c += interp, --pi) {
    const float2 cc = sc[c], p = *pi;
    const float pr = cc.x * p.x - cc.y * p.y;     // complex·complex, math.h:40-43
    const float pq = cc.x * p.y + cc.y * p.x;
    xr = xr + pr;
    xi = xi + pq;
  }
  out[o] = make_float2(xr, xi);
}

// simple_agc (sdr.h:253-273).  (1) per-chunk mean power, summed sample by sample like the reference;
__global__ __launch_bounds__(64) void k_agc_power(const float2 *in, unsigned long long nchunks, float *amp2) {
  const unsigned long long c = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= nchunks) return;
  const float2 *p = in + c * 128;
  float a = 0.f;
  for (int i = 0; i < 128; ++i) a += p[i].x * p[i].x + p[i].y * p[i].y;
  amp2[c] = a / 128;
}
// (2) the EMA over the chunks → gain per chunk.  The recurrence est ← est·(1−bw) + amp2·bw is a sequential float chain (its
// rounding sequence is the reference's), so one wave walks it: 64 chunk powers are loaded at once (one per lane), the chain
// runs on wave-uniform values fed by v_readlane (≈ 4 dependent VALU ops per chunk instead of a global-load round trip), every
// lane keeps the estimate of its own chunk, and the 64 gains (sqrt, divide) are computed and stored in parallel.
__global__ __launch_bounds__(64) void k_agc_gains(float *amp2_gain, unsigned long long nchunks, float *estimated, float out_rms, float bw) {
  const unsigned lane = threadIdx.x;
  float est = *estimated;
  const float keep = 1 - bw;
  float nxt = lane < nchunks ? amp2_gain[lane] : 0.f;
  for (unsigned long long c0 = 0; c0 < nchunks; c0 += 64) {
    const float cur = nxt;
    const unsigned long long cn = c0 + 64 + lane;
    nxt = cn < nchunks ? amp2_gain[cn] : 0.f;          // next block's powers are in flight during this block's chain
    const unsigned m = (unsigned)(nchunks - c0 < 64 ? nchunks - c0 : 64);
    float mine = 0.f;
    const float curbw = cur * bw;                      // the second product of every step, all 64 at once
    if (est != 0.f) {
      // est·keep + a·bw with est > 0 and a ≥ 0 cannot become 0 again: no "first chunk" test inside the chain
      for (unsigned i = 0; i < m; ++i) {
        est = est * keep + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(curbw), i));
        if (lane == i) mine = est;
      }
    } else {
      for (unsigned i = 0; i < m; ++i) {
        const float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cur), i));
        if (!est) est = a;
        est = est * keep + a * bw;
        if (lane == i) mine = est;
      }
    }
    if (lane < m) amp2_gain[c0 + lane] = mine ? out_rms / __builtin_sqrtf(mine) : 0.f;
  }
  if (lane == 0) *estimated = est;
}
// (3) out = in · gain[chunk]
__global__ __launch_bounds__(256) void k_agc_apply(const float2 *in, unsigned long long n, const float *gain, float2 *out) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float g = gain[i >> 7];
    out[i] = make_float2(in[i].x * g, in[i].y * g);
  }
}

static unsigned grid_for(lsdr_ctx *c, unsigned long long n, unsigned per = 256) {
  unsigned long long b = (n + per - 1) / per, cap = (unsigned long long)c->num_cu * 16;
  return (unsigned)(b < cap ? (b ? b : 1) : cap);
}

}  // namespace

struct lsdr_randomizer { lsdr_ctx *ctx; unsigned pos; unsigned char *d_pattern; };
struct lsdr_convol { lsdr_ctx *ctx; int bits_in, bits_out, bps; unsigned short polys[8]; unsigned short hist; };
struct lsdr_fir_resampler { lsdr_ctx *ctx; unsigned ncoeffs, interp; std::vector<float> coeffs; float2 *d_sc; float current_freq; };
struct lsdr_simple_agc { lsdr_ctx *ctx; float out_rms, bw; float *d_est, *d_gain; size_t gain_cap; };
// Read-only tables shared by every context of a device, created on first use.  Contexts may be driven from different threads
// (include/lsdr_hip.h): creation is serialised, and a table pointer is published only after its (synchronous) upload.
static std::mutex g_tx_tables_mutex;
static gf_tab *tx_gf_tables(lsdr_ctx *c) {
  static gf_tab *d_tab[64] = {nullptr};
  if (c->device < 0 || c->device >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_tx_tables_mutex);
  if (!d_tab[c->device]) {
    gf_tab g;
    lsdr_rs_tables(g.exp, g.log, g.G);
    gf_tab *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(gf_tab)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, &g, sizeof(g), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    d_tab[c->device] = d;
  }
  return d_tab[c->device];
}

extern "C" {

int lsdr_randomizer_create(lsdr_ctx *c, lsdr_randomizer **out) {
  LSDR_ARG(c && out);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_randomizer *r = new lsdr_randomizer();
  r->ctx = c; r->pos = 0;
  unsigned char pat[1504];
  lsdr_derandomizer_pattern(pat);      // same precompute_pattern() (dvb.h:1074-1087 ≡ :1116-1129)
  LSDR_HIP(hipMalloc((void **)&r->d_pattern, 1504));
  LSDR_HIP(hipMemcpy(r->d_pattern, pat, 1504, hipMemcpyHostToDevice));
  *out = r;
  return LSDR_OK;
}
void lsdr_randomizer_destroy(lsdr_randomizer *r) { if (r) { (void)hipStreamSynchronize(r->ctx->stream); (void)hipFree(r->d_pattern); delete r; } }
int lsdr_randomizer_run(lsdr_randomizer *r, const uint8_t *in_packets, size_t n_packets, uint8_t *out_packets, size_t cap_packets,
                        size_t *consumed, size_t *produced) {
  LSDR_ARG(r && consumed && produced);
  size_t n = n_packets < cap_packets ? n_packets : cap_packets;
  *consumed = *produced = n;
  if (!n) return LSDR_OK;
  LSDR_ARG(in_packets && out_packets);
  hipLaunchKernelGGL(k_randomize, dim3(grid_for(r->ctx, n * kTS)), dim3(256), 0, r->ctx->stream, in_packets,
                     (unsigned long long)n * kTS, (const unsigned char *)r->d_pattern, r->pos, out_packets);
  LSDR_HIP(hipGetLastError());
  r->pos = (unsigned)((r->pos + n * kTS) % 1504);
  return LSDR_OK;
}

int lsdr_rs_encoder_run(lsdr_ctx *c, const uint8_t *in_packets, size_t n_packets, uint8_t *out_packets, size_t cap_packets,
                        size_t *consumed, size_t *produced) {
  LSDR_ARG(c && consumed && produced);
  size_t n = n_packets < cap_packets ? n_packets : cap_packets;
  *consumed = *produced = n;
  if (!n) return LSDR_OK;
  LSDR_ARG(in_packets && out_packets);
  LSDR_HIP(hipSetDevice(c->device));
  gf_tab *t = tx_gf_tables(c);
  if (!t) { lsdr_set_error("rs_encoder: cannot create the GF(256) tables"); return LSDR_E_HIP; }
  hipLaunchKernelGGL(k_rs_encode, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, in_packets, (unsigned long long)n,
                     (const gf_tab *)t, out_packets);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

int lsdr_interleaver_run(lsdr_ctx *c, const uint8_t *in_packets, size_t n_packets, uint8_t *out_bytes, size_t cap_bytes,
                         size_t *consumed_packets, size_t *produced_bytes) {
  LSDR_ARG(c && consumed_packets && produced_bytes);
  *consumed_packets = 0; *produced_bytes = 0;
  if (n_packets < 12) return LSDR_OK;                       // while in.readable() >= 12 … (dvb.h:906)
  size_t n = n_packets - 11;
  if (n > cap_bytes / kRS) n = cap_bytes / kRS;
  if (!n) return LSDR_OK;
  LSDR_ARG(in_packets && out_bytes);
  hipLaunchKernelGGL(k_interleave, dim3(grid_for(c, n * kRS)), dim3(256), 0, c->stream, in_packets, (unsigned long long)n, out_bytes);
  LSDR_HIP(hipGetLastError());
  *consumed_packets = n; *produced_bytes = n * kRS;
  return LSDR_OK;
}

int lsdr_fec_spec(int rate, int *bits_in, int *bits_out, uint16_t polys_host[8]) {   // fec_specs, dvb.h:553-565
  static const unsigned short G1 = 0171, G2 = 0133;
  static const unsigned short p12[] = {G1, G2}, p23[] = {G1, G2, (unsigned short)(G2 << 1)},
                       p46[] = {G1, G2, (unsigned short)(G2 << 1), (unsigned short)(G1 << 2), (unsigned short)(G2 << 2), (unsigned short)(G2 << 3)},
                       p34[] = {G1, G2, (unsigned short)(G2 << 1), (unsigned short)(G1 << 2)},
                       p45[] = {G1, G2, (unsigned short)(G2 << 1), (unsigned short)(G1 << 2), (unsigned short)(G1 << 3)},
                       p56[] = {G1, G2, (unsigned short)(G2 << 1), (unsigned short)(G1 << 2), (unsigned short)(G2 << 3), (unsigned short)(G1 << 4)},
                       p78[] = {G1, G2, (unsigned short)(G2 << 1), (unsigned short)(G2 << 2), (unsigned short)(G2 << 3), (unsigned short)(G1 << 4),
                                (unsigned short)(G2 << 5), (unsigned short)(G1 << 6)};
  const unsigned short *p = nullptr;
  int bi = 0, bo = 0;
  switch (rate) {
    case LSDR_FEC12: bi = 1; bo = 2; p = p12; break;
    case LSDR_FEC23: bi = 2; bo = 3; p = p23; break;
    case LSDR_FEC46: bi = 4; bo = 6; p = p46; break;
    case LSDR_FEC34: bi = 3; bo = 4; p = p34; break;
    case LSDR_FEC56: bi = 5; bo = 6; p = p56; break;
    case LSDR_FEC78: bi = 7; bo = 8; p = p78; break;
    case LSDR_FEC45: bi = 4; bo = 5; p = p45; break;
    default: lsdr_set_error("fec_spec: Unexpected FEC"); return LSDR_E_ARG;
  }
  if (bits_in) *bits_in = bi;
  if (bits_out) *bits_out = bo;
  if (polys_host) for (int i = 0; i < 8; ++i) polys_host[i] = i < bo ? p[i] : 0;
  return LSDR_OK;
}

int lsdr_convol_create(lsdr_ctx *c, int rate, int bits_per_symbol, lsdr_convol **out) {
  LSDR_ARG(c && out && bits_per_symbol >= 1 && bits_per_symbol <= 8);
  lsdr_convol *v = new lsdr_convol();
  v->ctx = c; v->bps = bits_per_symbol; v->hist = 0;
  uint16_t p[8];
  if (lsdr_fec_spec(rate, &v->bits_in, &v->bits_out, p) != LSDR_OK) { delete v; lsdr_set_error("dvb_convol: Unexpected FEC"); return LSDR_E_ARG; }
  if (v->bits_out % v->bps) { delete v; lsdr_set_error("dvb_convol: Code rate not suitable for this constellation"); return LSDR_E_ARG; }
  for (int i = 0; i < 8; ++i) v->polys[i] = p[i];
  *out = v;
  return LSDR_OK;
}
void lsdr_convol_destroy(lsdr_convol *v) { delete v; }
int lsdr_convol_run(lsdr_convol *v, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *consumed, size_t *produced) {
  LSDR_ARG(v && consumed && produced);
  *consumed = 0; *produced = 0;
  long long count = (long long)n_in;                                        // dvb.h:587-591
  const long long lim = (long long)(cap_out * (size_t)v->bps / (size_t)v->bits_out * (size_t)v->bits_in / 8);
  if (lim < count) count = lim;
  count = (count / v->bits_in) * v->bits_in;
  if (count <= 0) return LSDR_OK;
  LSDR_ARG(in && out);
  const size_t nsym = (size_t)count * 8 / v->bits_in * v->bits_out / v->bps;
  convol_args a;
  a.in = in; a.nbytes = (unsigned long long)count; a.out = out; a.nsym = nsym;
  a.bits_in = v->bits_in; a.bits_out = v->bits_out; a.bps = v->bps;
  for (int i = 0; i < 8; ++i) a.polys[i] = v->polys[i];
  a.hist0 = v->hist;
  hipLaunchKernelGGL(k_convol, dim3((unsigned)((nsym + 255) / 256)), dim3(256), 0, v->ctx->stream, a);
  LSDR_HIP(hipGetLastError());
  // carried history = the last 16 input bits (fetch the last 2 bytes; older bits come from the previous history)
  unsigned char tail[2] = {0, 0};
  const size_t nb = count >= 2 ? 2 : 1;
  LSDR_HIP(hipMemcpyAsync(tail + (2 - nb), in + count - nb, nb, hipMemcpyDeviceToHost, v->ctx->stream));
  LSDR_HIP(hipStreamSynchronize(v->ctx->stream));
  if (count >= 2) {
    unsigned h = 0;   // bit 15 = newest = last bit of the last byte
    const unsigned w = ((unsigned)tail[0] << 8) | tail[1];     // input bits in stream order, MSB first
    for (int k = 0; k < 16; ++k) h |= ((w >> k) & 1u) << (15 - k);
    v->hist = (unsigned short)h;
  } else {
    unsigned h = (unsigned)v->hist >> 8;                         // 8 new bits push the old ones down
    for (int k = 0; k < 8; ++k) h |= (((unsigned)tail[1] >> k) & 1u) << (15 - k);
    v->hist = (unsigned short)h;
  }
  *consumed = (size_t)count; *produced = nsym;
  return LSDR_OK;
}

int lsdr_cstln_transmitter_run(lsdr_ctx *c, int cstln, int rate, const uint8_t *sym, size_t n, lsdr_cf32 *out) {
  LSDR_ARG(c && (n == 0 || (sym && out)));
  if (!n) return LSDR_OK;
  LSDR_HIP(hipSetDevice(c->device));
  static float2 *d_pts[64][16][16] = {};   // [device][constellation][code rate], created on first use
  LSDR_ARG(c->device >= 0 && c->device < 64 && cstln >= 0 && cstln < 16 && rate >= 0 && rate < 16);
  float2 *d = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_tx_tables_mutex);
    float2 *&slot = d_pts[c->device][cstln][rate];
    if (!slot) {
      lsdr::cstln_tables tab;
      if (lsdr::build_cstln(cstln, rate, tab) < 0) { lsdr_set_error("cstln_transmitter: constellation/code rate not supported"); return LSDR_E_ARG; }
      std::vector<float2> pts(256, make_float2(0.f, 0.f));
      for (int s = 0; s < tab.nsymbols; ++s) pts[s] = make_float2((float)(0 + tab.symbols[s][0]), (float)(0 + tab.symbols[s][1]));   // Zout + cp (sdr.h:1213-1214)
      float2 *nd = nullptr;
      LSDR_HIP(hipMalloc((void **)&nd, 256 * sizeof(float2)));
      if (hipMemcpy(nd, pts.data(), 256 * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(nd); LSDR_HIP(hipErrorUnknown); }
      slot = nd;
    }
    d = slot;
  }
  hipLaunchKernelGGL(k_cstln_map, dim3(grid_for(c, n)), dim3(256), 0, c->stream, sym, (unsigned long long)n, (const float2 *)d, (float2 *)out);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

int lsdr_fir_resampler_create(lsdr_ctx *c, unsigned ncoeffs, const float *coeffs_host, unsigned interp, lsdr_fir_resampler **out) {
  LSDR_ARG(c && out && ncoeffs >= 1 && coeffs_host && interp >= 1);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_fir_resampler *f = new lsdr_fir_resampler();
  f->ctx = c; f->ncoeffs = ncoeffs; f->interp = interp;
  f->coeffs.assign(coeffs_host, coeffs_host + ncoeffs);
  LSDR_HIP(hipMalloc((void **)&f->d_sc, ncoeffs * sizeof(float2)));
  *out = f;
  return lsdr_fir_resampler_set_freq(f, 0.f);   // ctor ends with set_freq(0), dsp.h:303
}
void lsdr_fir_resampler_destroy(lsdr_fir_resampler *f) { if (f) { (void)hipStreamSynchronize(f->ctx->stream); (void)hipFree(f->d_sc); delete f; } }
int lsdr_fir_resampler_set_freq(lsdr_fir_resampler *f, float freq) {   // dsp.h:351-360
  LSDR_ARG(f);
  std::vector<float2> sc(f->ncoeffs);
  for (unsigned i = 0; i < f->ncoeffs; ++i) {
    const float a = 2 * M_PI * freq * i;
    const float cs = cosf(a), sn = sinf(a);
    sc[i] = make_float2(f->coeffs[i] * cs, f->coeffs[i] * sn);
  }
  LSDR_HIP(hipStreamSynchronize(f->ctx->stream));
  LSDR_HIP(hipMemcpy(f->d_sc, sc.data(), sc.size() * sizeof(float2), hipMemcpyHostToDevice));
  f->current_freq = freq;
  return LSDR_OK;
}
int lsdr_fir_resampler_run(lsdr_fir_resampler *f, const lsdr_cf32 *in, size_t n_in, lsdr_cf32 *out, size_t cap_out, size_t *consumed,
                           size_t *produced) {
  LSDR_ARG(f && consumed && produced);
  *consumed = 0; *produced = 0;
  if (n_in < f->ncoeffs) return LSDR_OK;                                    // dsp.h:307
  if (n_in * f->interp < f->ncoeffs) return LSDR_OK;                        // dsp.h:318
  size_t count = (n_in * f->interp - f->ncoeffs) / f->interp;
  if (count > cap_out / f->interp) count = cap_out / f->interp;
  if (!count) return LSDR_OK;
  LSDR_ARG(in && out);
  const unsigned latency = (f->ncoeffs + f->interp) / f->interp;
  const unsigned long long nout = (unsigned long long)count * f->interp;
  hipLaunchKernelGGL(k_fir_resample, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, f->ctx->stream, (const float2 *)in,
                     (const float2 *)f->d_sc, f->ncoeffs, f->interp, latency, nout, (float2 *)out);
  LSDR_HIP(hipGetLastError());
  *consumed = count; *produced = (size_t)nout;
  return LSDR_OK;
}

int lsdr_simple_agc_create(lsdr_ctx *c, float out_rms, float bw, lsdr_simple_agc **out) {
  LSDR_ARG(c && out);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_simple_agc *a = new lsdr_simple_agc();
  a->ctx = c; a->out_rms = out_rms; a->bw = bw; a->d_gain = nullptr; a->gain_cap = 0;
  LSDR_HIP(hipMalloc((void **)&a->d_est, sizeof(float)));
  LSDR_HIP(hipMemset(a->d_est, 0, sizeof(float)));
  *out = a;
  return LSDR_OK;
}
void lsdr_simple_agc_destroy(lsdr_simple_agc *a) { if (a) { (void)hipStreamSynchronize(a->ctx->stream); (void)hipFree(a->d_est); (void)hipFree(a->d_gain); delete a; } }
int lsdr_simple_agc_set(lsdr_simple_agc *a, float out_rms, float bw) { LSDR_ARG(a); a->out_rms = out_rms; a->bw = bw; return LSDR_OK; }
int lsdr_simple_agc_run(lsdr_simple_agc *a, const lsdr_cf32 *in, size_t n_in, lsdr_cf32 *out, size_t cap_out, size_t *consumed,
                        size_t *produced) {
  LSDR_ARG(a && consumed && produced);
  size_t chunks = n_in / 128;
  if (chunks > cap_out / 128) chunks = cap_out / 128;
  *consumed = *produced = chunks * 128;
  if (!chunks) return LSDR_OK;
  LSDR_ARG(in && out);
  lsdr_ctx *c = a->ctx;
  if (a->gain_cap < chunks) {
    LSDR_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(a->d_gain);
    LSDR_HIP(hipMalloc((void **)&a->d_gain, chunks * sizeof(float)));
    a->gain_cap = chunks;
  }
  hipLaunchKernelGGL(k_agc_power, dim3((unsigned)((chunks + 63) / 64)), dim3(64), 0, c->stream, (const float2 *)in, (unsigned long long)chunks, a->d_gain);
  hipLaunchKernelGGL(k_agc_gains, dim3(1), dim3(64), 0, c->stream, a->d_gain, (unsigned long long)chunks, a->d_est, a->out_rms, a->bw);
  hipLaunchKernelGGL(k_agc_apply, dim3(grid_for(c, chunks * 128)), dim3(256), 0, c->stream, (const float2 *)in, (unsigned long long)chunks * 128,
                     (const float *)a->d_gain, (float2 *)out);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

}  // extern "C"
