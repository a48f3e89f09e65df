// leansdr_amd/csrc/cstln_receiver.hip — symbol timing + carrier recovery + slicer.
//
// Replaces cstln_receiver<f32>::run and the three sampler_interface<f32>
// implementations (sdr.h:589-938).  The loop is a per-symbol decision-feedback
// recurrence (PLL + Mueller&Müller timing + AGC): strictly sequential inside
// one stream.  Two execution modes share one device routine
// (rx_chunk, the body of the reference's 128-sample chunk):
//
//  LSDR_RX_SERIAL  one wavefront; lane 0 runs the recurrence with the
//                  reference's exact float operation order (this file is
//                  compiled -ffp-contract=off), the other 63 lanes stage the
//                  next chunk of samples into LDS with coalesced loads.
//                  Bit-exact soft symbols and loop state → parity anchor.
//  LSDR_RX_TILED   the stream is cut into tiles of `tile_len` samples; one lane
//                  per tile starts `tile_warmup` samples early from the
//                  current tracking state (freqw, AGC) with mu=phase=0, lets the
//                  loops converge, then emits its tile.  Seams are reconciled
//                  afterwards (quadrant of the carrier phase, symbol count) and
//                  the tiles are compacted.  Throughput mode: soft symbols agree
//                  with the serial result within the tolerance stated in
//                  tests/test_rx_tiled.py; see DESIGN.md §receiver.
//
// Look-up tables (trig16 512 KiB, constellation 512 KiB packed) are built on the
// host with the reference's libm calls (host_tables.cpp) and live in HBM; the hot
// region of both is L2-resident.  The constellation entry is packed to 8 bytes
// {cost, symbol, 0, phase_error, point.re, point.im} so one gather serves the
// slicer, the PLL and the timing detector.
#include <cmath>
#include "lsdr_internal.h"

namespace {

constexpr int kChunk = 128;   // cstln_receiver::chunk_size, sdr.h:706
constexpr int kRrcLdsTaps = 1024;   // fir_sampler taps the tolerance tiles keep in LDS (8 KiB per wavefront); longer filters read the table in HBM

#ifdef LSDR_RX_TRACE   // instrumented builds only (tools/): per-phase cycle sums of the symbol body
__device__ unsigned long long g_rx_probe[8];
#define LSDR_RXP(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); unsigned long long t__ = __builtin_amdgcn_s_memtime(); \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  if (i > 0) rxp_acc[i] += t__ - rxp_t; else if (rxp_t) rxp_acc[4] += t__ - rxp_t; \
  if (i == 3) { rxp_acc[0] += 1; unsigned long long u__ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); rxp_acc[5] += u__ - t__; t__ = u__; } \
  rxp_t = t__; }
#define LSDR_RXP_FLUSH { if (threadIdx.x == 0 && blockIdx.x == (gridDim.x > 1 ? 1u : 0u)) for (int k__ = 0; k__ < 6; ++k__) atomicAdd(&g_rx_probe[k__], rxp_acc[k__]); }
#else
#define LSDR_RXP(i)
#define LSDR_RXP_FLUSH
#endif
constexpr float kCstlnAmp = 75.0f;  // sdr.h:297

struct __attribute__((aligned(8))) lut_entry {
  int16_t cost;
  uint8_t symbol;
  uint8_t zero;
  int16_t phase_error;
  int8_t pt_re, pt_im;
};

struct rx_state_dev {            // sdr.h:923-935 + sampler state
  float mu, phase, freqw, agc_gain, est_insp, est_sp, est_ep;
  float min_freqw, max_freqw;
  float samp_freqw;              // linear_sampler::freqw (sdr.h:625)
  int update_freq_phase;         // fir_sampler::update_freq_phase (sdr.h:688)
  unsigned long long meas_count;
  float hist[12];
};

// One measurement instant (sdr.h:905-913).  Serial mode: the estimator values themselves.  Tiled mode: a tile only knows
// the affine map from "estimators at the start of the tile" to "estimators here" (est ↦ a·est + b, b in the est_* fields);
// k_rx_ema turns it into values once the maps of the preceding tiles have been composed.
struct rx_meas { float freqw, est_insp, est_sp, est_ep, a; unsigned tile; };

// The per-chunk estimators est_insp / est_sp / est_ep (sdr.h:867-889) are first-order EMAs with a constant pole: every
// chunk that produced a symbol applies est ↦ (1−kest)·est + kest·x.  A run of chunks is therefore an affine map, and maps
// compose associatively — the tiled receiver records one per tile and scans them (k_rx_ema) instead of letting every
// tile restart its own EMA from the carried value.
struct rx_ema_map { float a, bi, bs, be; };
__device__ __forceinline__ rx_ema_map ema_then(rx_ema_map f, rx_ema_map g) {   // g ∘ f (f first)
  rx_ema_map r; r.a = g.a * f.a; r.bi = g.a * f.bi + g.bi; r.bs = g.a * f.bs + g.bs; r.be = g.a * f.be + g.be;
  return r;
}

struct rx_consts {
  float omega, freq_alpha, freq_beta, gain_mu, kest;
  int allow_drift, nsymbols;
  unsigned long long meas_decimation;
  // fir sampler
  int ncoeffs, subsampling;
  // tolerance tiles: the first acq_syms symbols of a warm-up run the loops in an acquisition gear
  int acq_syms;
  float acq_alpha, acq_gain_mu;
};

struct rx_tables {
  const float2 *trig;            // [65536]
  const lut_entry *lut;          // [65536]
  const float *coeffs;           // fir sampler prototype
  float2 *shifted;               // fir sampler shifted coefficients
  const float2 *shifted_tol;     // tiled mode: the same table rebuilt from the carried freqw at the start of every run, for the
                                 // tolerance tiles (`shifted` follows the reference's refresh throttle and belongs to tile 0)
};

// trig16::expi(float), math.h:108-110
__device__ __forceinline__ unsigned trig_index(float a) { return (unsigned)(uint16_t)(int16_t)(int32_t)a; }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {  // math.h:40-43
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// ---- input sample streams --------------------------------------------------------------------------------------------
// The receiver reads either the reference's cf32 items or — LSDR_IN_CU8, the leandvb --u8 graph (leandvb.cc:211-217) with
// cconverter<u8,128,f32,0,1,1> fused into the load — cu8 items converted exactly as dsp.h:40-50 does it:
// out = 0 + ((int)in − 128)·1/1 in int, then → float (exact).  The converted cf32 stream never exists in HBM.
__device__ __forceinline__ float2 cu8_to_cf32(unsigned re, unsigned im) {
  return make_float2((float)((int)re - 128), (float)((int)im - 128));
}
struct in_cf32 {
  const float2 *p;
  __device__ __forceinline__ float2 operator[](long long i) const { return p[i]; }
  __device__ __forceinline__ in_cf32 operator+(long long k) const { in_cf32 r; r.p = p + k; return r; }
};
struct in_cu8 {
  const uchar2 *p;
  __device__ __forceinline__ float2 operator[](long long i) const { const uchar2 v = p[i]; return cu8_to_cf32(v.x, v.y); }
  __device__ __forceinline__ in_cu8 operator+(long long k) const { in_cu8 r; r.p = p + k; return r; }
};
template <int FMT> struct in_stream { typedef in_cf32 type; };
template <> struct in_stream<LSDR_IN_CU8> { typedef in_cu8 type; };
template <int FMT> __device__ __forceinline__ typename in_stream<FMT>::type in_make(const void *p) {
  typename in_stream<FMT>::type r; r.p = reinterpret_cast<decltype(r.p)>(p); return r;
}

// The tolerance tiles' look-ahead window: the samples the NEXT symbol can need (instant n' = n + ⌊mu + omega + mucorr⌋ is one
// of two adjacent positions, the linear sampler reads n' and n'+1), requested one symbol step ahead.
//   cf32: three 8-byte loads, positions wn … wn+2 (each clamped to the tile's last readable sample);
//   cu8:  ONE 8-byte load = four samples wn … wn+3 (unaligned dwordx2; the window start is clamped so that it never reaches
//         past the last readable sample), the pair is picked with a 64-bit shift and converted with v_cvt_f32_ubyte*.
template <int FMT> struct rx_window;
template <> struct rx_window<LSDR_IN_CF32> {
  // samples wn and wn + 1 (what one linear interpolation reads), fetched by ONE 16-byte load: a tile walks its own part of
  // the stream, so every load instruction of a wavefront touches 64 different cache lines — the texture-address unit's time,
  // which fir_filter's streaming loads share, goes with the number of load instructions, not with their width
  struct __attribute__((aligned(8))) pair { float ar, ai, br, bi; };
  float2 w0, w1; int wn;
  __device__ __forceinline__ void load(const in_cf32 &b, int i, int n_last) {
    const int k = i < n_last ? i : n_last - 1;           // (a tile span is ≥ 128 samples)
    const pair v = *reinterpret_cast<const pair *>(b.p + k);
    w1 = make_float2(v.br, v.bi); w0 = i < n_last ? make_float2(v.ar, v.ai) : w1; wn = i;   // beyond the span the last sample repeats
  }
  __device__ __forceinline__ bool covers(int n) const { return n == wn; }
  __device__ __forceinline__ void get(int, float2 &p0, float2 &p1) const { p0 = w0; p1 = w1; }
};
template <> struct rx_window<LSDR_IN_CU8> {
  unsigned long long w; int wn;
  __device__ __forceinline__ void load(const in_cu8 &b, int i, int n_last) {
    const int lim = n_last - 3;                            // (a tile span is ≥ 128 samples)
    wn = i < lim ? i : lim;
    __builtin_memcpy(&w, reinterpret_cast<const unsigned char *>(b.p) + 2 * (long long)wn, 8);
  }
  __device__ __forceinline__ bool covers(int n) const { return (unsigned)(n - wn) <= 2u; }
  __device__ __forceinline__ void get(int n, float2 &p0, float2 &p1) const {
    const unsigned v = (unsigned)(w >> (16 * (n - wn)));   // {re0, im0, re1, im1}
    p0 = cu8_to_cf32(v & 255u, (v >> 8) & 255u);
    p1 = cu8_to_cf32((v >> 16) & 255u, v >> 24);
  }
};

// window starting at byte offset `bo` (clamped to the row by the caller) of a lane's LDS row; s0 = tile-relative index of the row's first
// sample, delta = byte offset of that sample in the row (the stream's misalignment against the 16-byte pieces)
__device__ __forceinline__ void rx_window_lds(rx_window<LSDR_IN_CU8> &wd, const char *row, int bo, int delta, int s0) {
  const int d = bo >> 2;                                  // two dwords starting at the dword that holds the sample
  const unsigned *r32 = reinterpret_cast<const unsigned *>(row);
  const unsigned lo32 = r32[d], hi32 = r32[d + 1];
  wd.w = ((unsigned long long)hi32 << 32) | lo32; wd.wn = s0 + ((4 * d - delta) >> 1);
}
__device__ __forceinline__ void rx_window_lds(rx_window<LSDR_IN_CF32> &wd, const char *row, int bo, int delta, int s0) {
  const float2 *r64 = reinterpret_cast<const float2 *>(row + bo);
  wd.w0 = r64[0]; wd.w1 = r64[1]; wd.wn = s0 + ((bo - delta) >> 3);
}

__device__ __forceinline__ void lut_halve(float &I, float &Q) {   // the range-folding loop of sdr.h:470-476 alone
  while (__builtin_fmaxf(__builtin_fabsf(I + 0.5f), __builtin_fabsf(Q + 0.5f)) > 127.5f) {
    I *= 0.5f;
    Q *= 0.5f;
  }
}
__device__ __forceinline__ unsigned lut_index(float I, float Q) {
  // `I < -128 || I > 127` ⟺ `|I + 0.5| > 127.5`: the addition is exact wherever the comparison could be
  // affected (|I| in [64, 256): 0.5 is a multiple of ulp(I) and the sum does not leave that grid), and
  // rounding is monotonic elsewhere.  One max + one compare instead of four compares.
  while (__builtin_fmaxf(__builtin_fabsf(I + 0.5f), __builtin_fabsf(Q + 0.5f)) > 127.5f) {
    I *= 0.5f;
    Q *= 0.5f;
  }
  return ((unsigned)(int)I & 255u) * 256u + ((unsigned)(int)Q & 255u);
}

// fmodf(x, 65536): exact (scaling by 2^16 and truncation are exact, the final
// subtraction is exact because the true remainder is representable).
__device__ __forceinline__ float fmod65536(float x) {
  float q = __builtin_truncf(x * (1.0f / 65536.0f));
  return x - q * 65536.0f;
}

// ---- QPSK decisions by arithmetic (tolerance tiles only) ------------------------------------------------------------
// What the 512 KiB constellation table holds for QPSK (sdr.h:334-337 points (±53,±53); sdr.h:529-560 table), computed
// from the truncated coordinates instead of gathered: the gather is the only memory access on the per-symbol dependency
// chain of a tile, and an L2 round trip under fir_filter's streaming costs more than these ≈ 35 ALU operations.
//   symbol / cost: EXACT (integer arithmetic: nearest point by signs — ties go to '+' like the table's lowest-index rule —
//     second nearest = flip the coordinate of smaller magnitude, d2 = d1 + 4·53·min(|I|,|Q|), both clamped to 32767);
//   phase_error: atan2 by a degree-11 odd polynomial, within ±2 units of the table's (s32)((atan2f(Q,I) − atan2f(sym))·65536/2π)
//     (the table's unit is 2π/65536 rad; the loops multiply it by 0.04 and 0.0003).
// lsdr_rx_create checks both claims against the real table over all 65536 entries and keeps the gather if they fail.
struct qpsk_decision { int cost; unsigned symbol; int phase_error, pt_re, pt_im; };
__host__ __device__ __forceinline__ qpsk_decision qpsk_decide(int Ii, int Qi) {
  qpsk_decision d;
  const int a = Ii < 0 ? -Ii : Ii, b = Qi < 0 ? -Qi : Qi;
  const bool ni = Ii < 0, nq = Qi < 0;
  d.symbol = (ni ? 2u : 0u) | (nq ? 1u : 0u);
  d.pt_re = ni ? -53 : 53; d.pt_im = nq ? -53 : 53;
  const int da = a - 53, db = b - 53, d1 = da * da + db * db;
  int d2 = d1 + 212 * (a < b ? a : b);
  d2 = d2 > 32767 ? 32767 : d2;
  d.cost = d1 - d2;                                    // d1 ≤ 2·75² < 32767
  const float fa = (float)a, fb = (float)b;
  const float mx = fa > fb ? fa : fb, mn = fa > fb ? fb : fa;
#ifdef __HIP_DEVICE_COMPILE__
  const float t = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;
#else
  const float t = mx > 0.f ? mn / mx : 0.f;
#endif
  const float t2 = t * t;
  // atan(t), 0 ≤ t ≤ 1 (max error 2e-6 rad)
  float p = -0.0117212f;
  p = p * t2 + 0.05265332f;
  p = p * t2 - 0.11643287f;
  p = p * t2 + 0.19354346f;
  p = p * t2 - 0.33262347f;
  p = p * t2 + 0.99997726f;
  const float at = p * t;
  const float ang = fb > fa ? 1.57079633f - at : at;  // atan2(b, a), first quadrant
  const int pe = (int)((ang - 0.785398163f) * 10430.3784f);   // truncation toward zero, like the table's (s32) cast
  d.phase_error = ni != nq ? -pe : pe;
  return d;
}

// ---- table access policies -------------------------------------------------------------------------
// rx_chunk is latency-bound on its two table gathers (trig16, constellation LUT), so how they are
// fetched is chosen per kernel:
//  ld_vec      per-lane vector gathers (one 8-byte load each);
//  ld_uniform  exactly one lane is active (the exact serial kernel): the index is made wave-uniform and
//              the entry comes through the scalar cache (s_load via the constant address space) — about
//              half the latency of a vector load that misses the per-CU L1;
//  ld_hwtrig   tolerance-mode tiles: expi() of the same 16-bit quantised angle from v_cos/v_sin_f32
//              (argument in revolutions) instead of the 512 KiB table — |error| ~1e-6, far below the
//              table's own angle quantisation (1e-4 rad); not bit-identical, never used where exactness
//              is promised.
typedef float rx_v2f __attribute__((ext_vector_type(2)));
typedef unsigned rx_v2u __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) rx_v2f *rx_cptr_f2;
typedef const __attribute__((address_space(4))) rx_v2u *rx_cptr_u2;

__device__ __forceinline__ lut_entry lut_unpack(unsigned lo, unsigned hi) {
  lut_entry e;
  e.cost = (int16_t)(lo & 0xffffu); e.symbol = (uint8_t)((lo >> 16) & 0xffu); e.zero = 0;
  e.phase_error = (int16_t)(hi & 0xffffu); e.pt_re = (int8_t)((hi >> 16) & 0xffu); e.pt_im = (int8_t)(hi >> 24);
  return e;
}
struct ld_vec {
  static __device__ __forceinline__ float2 expi(const rx_tables &T, unsigned i) { return T.trig[i]; }
  static __device__ __forceinline__ lut_entry lut(const rx_tables &T, unsigned i) {
    const uint2 raw = *reinterpret_cast<const uint2 *>(T.lut + i);
    return lut_unpack(raw.x, raw.y);
  }
};
struct ld_uniform {
  static __device__ __forceinline__ float2 expi(const rx_tables &T, unsigned i) {
    const unsigned si = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
    const rx_v2f v = ((rx_cptr_f2)T.trig)[si];
    return make_float2(v.x, v.y);
  }
  static __device__ __forceinline__ lut_entry lut(const rx_tables &T, unsigned i) {
    const unsigned si = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
    const rx_v2u raw = ((rx_cptr_u2)T.lut)[si];
    return lut_unpack(raw.x, raw.y);
  }
};
struct ld_hwtrig {
  static __device__ __forceinline__ float2 expi(const rx_tables &, unsigned i) {
    const float rev = (float)i * (1.0f / 65536.0f);
    return make_float2(__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev));
  }
  static __device__ __forceinline__ lut_entry lut(const rx_tables &T, unsigned i) { return ld_vec::lut(T, i); }
};

// sampler_interface::interp — SAMP: 0 nearest (sdr.h:602-604), 1 linear
// (sdr.h:614-623), 2 fir (sdr.h:646-665).
template <int SAMP, typename LD, typename SamplePtr>
__device__ __forceinline__ float2 interp(const rx_tables &T, const rx_consts &C, const rx_state_dev &s,
                                         SamplePtr pin, float mu, float phase) {
  if (SAMP == 0) return cmul(pin[0], LD::expi(T, trig_index(-phase)));
  if (SAMP == 1) {
    float2 s0 = cmul(pin[0], LD::expi(T, trig_index(-phase)));
    float2 s1 = cmul(pin[1], LD::expi(T, trig_index(-(phase + s.samp_freqw))));
    float k0 = 1 - mu;
    return make_float2(s0.x * k0 + s1.x * mu, s0.y * k0 + s1.y * mu);
  }
  float2 acc = make_float2(0.f, 0.f);
  const int S = C.subsampling, N = C.ncoeffs;
  int k = 0;
  for (int pc = (int)((1 - mu) * S); pc < N; pc += S, ++k) {
    float2 t = cmul(T.shifted[pc], pin[k]);
    acc.x += t.x;
    acc.y += t.y;
  }
  return cmul(LD::expi(T, trig_index(-phase)), acc);
}

// One 128-sample chunk of cstln_receiver::run (sdr.h:790-913), run by ONE lane.
// Emits symbols through `emit(softsymbol)`; returns the number emitted.
// last_s / last_sg / had_symbol feed the per-chunk estimators.
// WIN (tolerance tiles only): freqw is kept within [f_lo, f_hi] around the carried value — a tile that starts from
// scratch next to the PLL's unstable equilibrium can otherwise pump the frequency integrator into a false lock that
// outlasts the tile (observed with the integer receiver of hs.hip; same loop structure here).
template <int SAMP, typename LD, bool WIN = false, typename SamplePtr, typename Emit>
__device__ __forceinline__ int rx_chunk(const rx_tables &T, const rx_consts &C, rx_state_dev &s, SamplePtr pin,
                                        Emit emit, float2 *cstln_out_slot, bool *wrote_cstln, float f_lo = 0.f,
                                        float f_hi = 0.f) {
  float mu = s.mu, phase = s.phase, freqw = s.freqw;
  const float agc_gain = s.agc_gain;
  float2 sg = make_float2(0.f, 0.f), sv = make_float2(0.f, 0.f);
  int pt_re = 0, pt_im = 0;
  bool had = false;
  int nsym = 0;
  float h0pr = s.hist[0], h0pi = s.hist[1], h0cr = s.hist[2], h0ci = s.hist[3];
  float h1pr = s.hist[4], h1pi = s.hist[5], h1cr = s.hist[6], h1ci = s.hist[7];
  float h2pr = s.hist[8], h2pi = s.hist[9], h2cr = s.hist[10], h2ci = s.hist[11];

  // The reference walks the chunk sample by sample: `if (mu<1) {symbol}; ++pin; --mu;
  // phase += freqw` (sdr.h:800-847).  Here the samples without a symbol are consumed by
  // the short inner loop (the same two float operations per sample, in the same
  // order), so the expensive symbol body is executed once per loop trip by every lane
  // of a wavefront together — lanes of the tiled kernel hit their symbol instants at
  // different sample indices and would otherwise serialise.  Bit-identical results.
#ifdef LSDR_RX_TRACE
  unsigned long long rxp_t = 0, rxp_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  // Sample steps without a symbol, `while (!(mu < 1) && n < kChunk) { mu -= 1; phase += freqw; ++n; }`:
  // k = number of trips.  For 1 ≤ mu < 2^24 every `mu - 1` is exact, so k = trunc(mu) (capped by the
  // chunk end) and `mu - k` is the same float as k successive subtractions.  The k roundings of
  // `phase += freqw` are kept: a wave-uniform number (kmax ≥ any k the loop dynamics can produce) of
  // add+select steps, no data-dependent branch.  Anything unusual (NaN, huge mu, k > kmax) takes the
  // reference's loop literally.
  const int kmax = (int)C.omega + 2;
  int n = 0;
  auto skip = [&]() {
    const int rem = kChunk - n;
    if (mu < 16777216.0f) {
      int k = mu < 1 ? 0 : (int)mu;
      k = k < rem ? k : rem;
      mu = mu - (float)k;
      n += k;
      float p = phase;
      for (int i = 0; i < kmax; ++i) {
        p += freqw;
        phase = i < k ? p : phase;
      }
      for (int i = kmax; i < k; ++i) phase += freqw;
    } else {
      while (!(mu < 1) && n < kChunk) {
        mu = mu - 1;
        phase += freqw;
        ++n;
      }
    }
  };
  skip();
  while (n < kChunk) {
    {
      LSDR_RXP(0)
      sg = interp<SAMP, LD>(T, C, s, pin + n, mu, phase);
      sv = make_float2(sg.x * agc_gain, sg.y * agc_gain);
      LSDR_RXP(1)
      const lut_entry e = LD::lut(T, lut_index(sv.x, sv.y));
      LSDR_RXP(2)
      lsdr_softsymbol ss;
      ss.cost = e.cost; ss.symbol = e.symbol; ss.pad = 0;
      emit(ss);
      ++nsym;
      // PLL, sdr.h:814-815
      phase += e.phase_error * C.freq_alpha;
      freqw += e.phase_error * C.freq_beta;
      if (WIN) { freqw = freqw < f_lo ? f_lo : freqw; freqw = freqw > f_hi ? f_hi : freqw; }
      // Modified Mueller & Müller, sdr.h:822-840
      h2pr = h1pr; h2pi = h1pi; h2cr = h1cr; h2ci = h1ci;
      h1pr = h0pr; h1pi = h0pi; h1cr = h0cr; h1ci = h0ci;
      h0pr = sv.x; h0pi = sv.y;
      pt_re = e.pt_re; pt_im = e.pt_im;
      had = true;
      h0cr = (float)pt_re; h0ci = (float)pt_im;
      float muerr = ((h0pr - h2pr) * h1cr + (h0pi - h2pi) * h1ci) -
                    ((h0cr - h2cr) * h1pr + (h0ci - h2ci) * h1pi);
      float mucorr = muerr * C.gain_mu;
      const float max_mucorr = 0.1f;
      if (mucorr < -max_mucorr) mucorr = -max_mucorr;
      if (mucorr > max_mucorr) mucorr = max_mucorr;
      mu += mucorr;
      mu += C.omega;
      LSDR_RXP(3)
    }
    mu = mu - 1;
    phase += freqw;
    ++n;
    skip();
  }
  LSDR_RXP_FLUSH
  phase = fmod65536(phase);  // sdr.h:855

  float est_insp = s.est_insp, est_sp = s.est_sp, est_ep = s.est_ep, agc = s.agc_gain;
  if (had) {
    if (cstln_out_slot) { *cstln_out_slot = sv; }
    *wrote_cstln = true;
    float insp = sg.x * sg.x + sg.y * sg.y;          // sdr.h:867-870
    est_insp = insp * C.kest + est_insp * (1 - C.kest);
    if (est_insp) agc = kCstlnAmp / __builtin_sqrtf(est_insp);
    float evr = sv.x - pt_re, evi = sv.y - pt_im;     // sdr.h:873-889
    float sig_power, ev_power;
    if (C.nsymbols == 2) {
      float sig_real = (float)((double)(pt_re + pt_im) * 0.707);
      float ev_real = (float)((double)(evr + evi) * 0.707);
      sig_power = sig_real * sig_real;
      ev_power = ev_real * ev_real;
    } else {
      sig_power = (float)(pt_re * pt_re + pt_im * pt_im);
      ev_power = evr * evr + evi * evi;
    }
    est_sp = sig_power * C.kest + est_sp * (1 - C.kest);
    est_ep = ev_power * C.kest + est_ep * (1 - C.kest);
  } else {
    *wrote_cstln = false;
  }
  if (!C.allow_drift) {                                // sdr.h:895-898
    if (freqw < s.min_freqw || freqw > s.max_freqw) freqw = (s.max_freqw + s.min_freqw) / 2;
  }
  s.mu = mu; s.phase = phase; s.freqw = freqw;
  s.est_insp = est_insp; s.est_sp = est_sp; s.est_ep = est_ep; s.agc_gain = agc;
  s.hist[0] = h0pr; s.hist[1] = h0pi; s.hist[2] = h0cr; s.hist[3] = h0ci;
  s.hist[4] = h1pr; s.hist[5] = h1pi; s.hist[6] = h1cr; s.hist[7] = h1ci;
  s.hist[8] = h2pr; s.hist[9] = h2pi; s.hist[10] = h2cr; s.hist[11] = h2ci;
  return nsym;
}

struct rx_serial_args {
  const void *in;                  // cf32 or cu8 items (FMT)
  unsigned long long n_in;
  lsdr_softsymbol *out;
  unsigned long long cap_out;
  rx_state_dev *state;
  rx_meas *meas; unsigned long long meas_cap;
  float2 *cstln; unsigned long long cstln_cap;
  unsigned long long *counters;   // [0] consumed, [1] produced, [2] n_meas, [3] n_cstln
  rx_consts C;
  rx_tables T;
  int readahead;
};

// One wavefront.  Lane 0 = the recurrence; all lanes = staging + coefficient refresh.
template <int SAMP, int FMT>
__global__ __launch_bounds__(64) void k_rx_serial(rx_serial_args a) {
  const typename in_stream<FMT>::type src = in_make<FMT>(a.in);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2 *buf = reinterpret_cast<float2 *>(smem_raw);     // [kChunk + readahead]
  __shared__ rx_state_dev st;
  __shared__ unsigned long long sh_nout, sh_nm, sh_nc;
  const int lane = threadIdx.x;
  if (lane == 0) { st = *a.state; sh_nout = 0; sh_nm = 0; sh_nc = 0; }
  __syncthreads();

  const unsigned long long max_meas = kChunk / a.C.meas_decimation + 1;
  const int span = kChunk + a.readahead;
  unsigned long long pos = 0;
  auto available = [&](unsigned long long p) { return a.n_in >= p && a.n_in - p >= (unsigned long long)span; };
  if (available(0))
    for (int k = lane; k < span; k += 64) buf[k] = src[k];
  __syncthreads();
  while (true) {
    // loop condition of sdr.h:783-788 (uniform: shared counters)
    if (!available(pos)) break;
    if (a.cap_out - sh_nout < kChunk) break;
    if (a.meas_cap - sh_nm < max_meas) break;
    if (a.cstln_cap - sh_nc < max_meas) break;

    // the next chunk's samples are requested now and land in registers while lane 0 works on this one
    const bool have_next = available(pos + kChunk);
    float2 pre[3];
    if (have_next) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (lane + 64 * q < span) pre[q] = src[pos + kChunk + lane + 64 * q];
    }

    // sampler->update_freq(freqw), sdr.h:790
    if (SAMP == 1) {
      if (lane == 0) st.samp_freqw = st.freqw;
    } else if (SAMP == 2) {
      // fir_sampler::update_freq (sdr.h:667-674); do_update_freq (sdr.h:676-680)
      // is a per-tap map, spread over the wave.
      int ph = st.update_freq_phase - 128;
      __syncthreads();
      if (ph <= 0) {
        float f = st.freqw / a.C.subsampling;
        const int N = a.C.ncoeffs;
        for (int i = lane; i < N; i += 64) {
          float2 e = a.T.trig[trig_index(-f * (i - N / 2))];
          float c = a.T.coeffs[i];
          a.T.shifted[i] = make_float2(e.x * c, e.y * c);
        }
        ph = N * 16;
        __threadfence_block();
      }
      if (lane == 0) st.update_freq_phase = ph;
    }
    __syncthreads();

    if (lane == 0) {
      lsdr_softsymbol *po = a.out + sh_nout;
      int cnt = 0;
      bool wrote = false;
      int n = rx_chunk<SAMP, ld_uniform>(a.T, a.C, st, (const float2 *)buf,
                                         [&](lsdr_softsymbol ss) { po[cnt++] = ss; },
                                         a.cstln ? a.cstln + sh_nc : nullptr, &wrote);
      sh_nout += n;
      if (wrote) sh_nc += 1;
      // measurements, sdr.h:905-913 (values finalised on the host with libm)
      st.meas_count += kChunk;
      while (st.meas_count >= a.C.meas_decimation) {
        st.meas_count -= a.C.meas_decimation;
        rx_meas m; m.freqw = st.freqw; m.est_insp = st.est_insp; m.est_sp = st.est_sp; m.est_ep = st.est_ep; m.a = 0.f; m.tile = 0u;
        a.meas[sh_nm++] = m;
      }
    }
    pos += kChunk;
    __syncthreads();
    if (have_next) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (lane + 64 * q < span) buf[lane + 64 * q] = pre[q];
      for (int k = lane + 192; k < span; k += 64) buf[k] = src[pos + k];
    }
    __syncthreads();
  }
  if (lane == 0) {
    *a.state = st;
    a.counters[0] = pos; a.counters[1] = sh_nout; a.counters[2] = sh_nm; a.counters[3] = sh_nc;
  }
}


// ---------------------------------------------------------------- LSDR_RX_TILED
// hooks of rx_tiling.h for this receiver
struct rx_state_dev;
__device__ __forceinline__ lsdr_softsymbol rx_relabel(lsdr_softsymbol v, const uint8_t *map) { v.symbol = map[v.symbol]; return v; }
__device__ __forceinline__ unsigned rx_symbol_of(lsdr_softsymbol v) { return v.symbol; }
__device__ __forceinline__ float fmod65536(float x);
__device__ __forceinline__ void rx_rotate_back(rx_state_dev *st, unsigned rot, float quad);
__device__ __forceinline__ float rx_freq_tap(const rx_state_dev *st) { return st->freqw / 65536; }   // refresh_freq_tap, sdr.h:919-921
#include "rx_tiling.h"
typedef rx_tile_info_t<lsdr_softsymbol> rx_tile_info;

__device__ __forceinline__ void rx_rotate_back(rx_state_dev *st, unsigned rot, float quad) {
  st->phase = fmod65536(st->phase - rot * quad);
}

struct rx_tiled_args {
  const void *in;                      // cf32 or cu8 items (FMT)
  unsigned long long total_chunks;     // chunks processed by this run
  unsigned first_chunks, tile_chunks, warm_chunks;
  unsigned n_tiles;
  unsigned lanes_per_wave;             // active lanes (tiles) per wavefront
  unsigned dbg;                        // measurement hooks (LSDR_RX_DBG; results are garbage): 1 no scattered symbol stores, 2 no window loads; 4: LSDR_RX_PRIO
  unsigned stage_stride;               // symbols reserved per tile in `stage`
  lsdr_softsymbol *stage;
  lsdr_softsymbol *wstage;             // [n_tiles][wstride]: symbols of each tile's last warm-up chunk (seam vote)
  unsigned wstride;
  rx_tile_info *info;
  unsigned *hstage;                    // LSDR_SYM_HARD2: packed body symbols (rx_tiling.h "hs2"), transposed: word w of tile j at
  unsigned long long hpitch;           //                 hstage[w·hpitch + j]
  rx_tile_info_h *hinfo;
  rx_ema_map *ema;                     // [n_tiles]: composition of the maps of the tiles BEFORE this one in its wavefront
  rx_ema_map *ema_wave;                // [n_waves]: composition of all tiles of a wavefront (k_rx_ema scans these)
  const rx_state_dev *state;           // carried state at the start of the run (read-only while tiles are running)
  rx_state_dev *state_next;            // end state of the last tile (k_rx_ema moves it into `state`)
  rx_meas *meas;                       // [n_meas] measurement slots (may be null)
  unsigned long long meas_base;        // meas_count at the start of the run
  float2 *cstln;                       // [total_chunks] last sampled point of every chunk (NaN: the chunk had no symbol); may be null
  rx_consts C;
  rx_tables T;
};

// measurement instants that fall into chunk c of the run (sdr.h:905-913: one per meas_decimation samples of the stream)
__device__ __forceinline__ void rx_tile_meas(const rx_tiled_args &a, unsigned long long c, unsigned tile, float freqw,
                                             const rx_ema_map &m) {
  const unsigned long long md = a.C.meas_decimation;
  const unsigned long long before = (a.meas_base + c * kChunk) / md, after = (a.meas_base + (c + 1) * kChunk) / md;
  const unsigned long long first_m = a.meas_base / md;
  for (unsigned long long q = before; q < after; ++q) {
    rx_meas mm; mm.freqw = freqw; mm.a = m.a; mm.est_insp = m.bi; mm.est_sp = m.bs; mm.est_ep = m.be; mm.tile = tile;
    a.meas[q - first_m] = mm;
  }
}

// Tile 0: continues exactly from the carried state (one lane, the reference's arithmetic, exact table look-ups).
template <int SAMP, int FMT, bool HARD>
__device__ __forceinline__ void rx_tile_exact(const rx_tiled_args &a) {
  const typename in_stream<FMT>::type src = in_make<FMT>(a.in);
  unsigned long long c1 = a.first_chunks;
  if (c1 > a.total_chunks) c1 = a.total_chunks;
  rx_state_dev s = *a.state;
  rx_tile_info ti;
  ti.has_pre = 0; ti.pre.cost = 0; ti.pre.symbol = 0; ti.pre.pad = 0; ti.n_warm = 0;
  ti.mu_begin = s.mu; ti.phase_begin = s.phase;
  lsdr_softsymbol *po = a.stage;
  unsigned cnt = 0;
  unsigned hacc = 0, htail = 0;          // HARD: word being filled, last 16 symbols
  auto emit = [&](lsdr_softsymbol ss) {
    if (HARD) {
      hacc = (hacc << 2) | (ss.symbol & 3u); htail = (htail << 2) | (ss.symbol & 3u);
      if ((++cnt & 15u) == 0) a.hstage[(unsigned long long)((cnt >> 4) - 1) * a.hpitch] = hacc;
    } else {
      po[cnt++] = ss;
    }
  };
  rx_ema_map m; m.a = 0.f; m.bi = s.est_insp; m.bs = s.est_sp; m.be = s.est_ep;   // constant map: the values are exact here
  for (unsigned long long c = 0; c < c1; ++c) {
    if (SAMP == 1) s.samp_freqw = s.freqw;
    if (SAMP == 2) {    // fir_sampler::update_freq (sdr.h:667-680): refresh the shifted taps once per N·16 samples
      int ph = s.update_freq_phase - 128;
      if (ph <= 0) {
        const float f = s.freqw / a.C.subsampling;
        const int N = a.C.ncoeffs;
        for (int i = 0; i < N; ++i) {
          const float2 e = ld_uniform::expi(a.T, trig_index(-f * (i - N / 2)));
          const float cc = a.T.coeffs[i];
          a.T.shifted[i] = make_float2(e.x * cc, e.y * cc);
        }
        ph = N * 16;
      }
      s.update_freq_phase = ph;
    }
    bool wrote;
    rx_chunk<SAMP, ld_uniform>(a.T, a.C, s, src + c * kChunk, emit, a.cstln ? a.cstln + c : nullptr, &wrote);
    if (a.cstln && !wrote) a.cstln[c] = make_float2(__builtin_nanf(""), __builtin_nanf(""));
    m.bi = s.est_insp; m.bs = s.est_sp; m.be = s.est_ep;
    if (a.meas) rx_tile_meas(a, c, 0u, s.freqw, m);
  }
  ti.mu_end = s.mu; ti.phase_end = s.phase; ti.count = cnt;
  if (HARD) {
    if (cnt & 15u) a.hstage[(unsigned long long)(cnt >> 4) * a.hpitch] = hacc << (2 * (16 - (cnt & 15u)));
    rx_tile_info_h th;
    th.mu_begin = ti.mu_begin; th.phase_begin = ti.phase_begin; th.mu_end = ti.mu_end; th.phase_end = ti.phase_end;
    th.count = cnt; th.has_pre = 0; th.n_warm = 0; th.warm_tail = 0; th.body_tail = htail;
    a.hinfo[0] = th;
  } else {
    a.info[0] = ti;
  }
  { rx_ema_map id; id.a = 1.f; id.bi = id.bs = id.be = 0.f; a.ema[0] = id; }
  a.ema_wave[0] = m;
  if (a.n_tiles == 1) {
    s.meas_count = (a.meas_base + a.total_chunks * kChunk) % a.C.meas_decimation;
    *a.state_next = s;
  }
}

// Tiles j ≥ 1 (tolerance mode): one lane per tile.  The lane starts `warm_chunks` early from the carried tracking state
// (freqw, AGC, estimators) with mu = phase = 0, lets the loops converge, then emits its tile.  Same chunk structure and
// operation order as rx_chunk, with three liberties that the tolerance contract pays for:
//   * expi() of the 16-bit quantised angle comes from v_cos/v_sin_f32 instead of the 512 KiB table;
//   * the sample-skipping steps advance the phase by k·freqw in one multiply-add instead of k additions;
//   * freqw is held inside a small window around the carried value (see rx_chunk's WIN).
// The per-symbol dependency chain is what bounds this kernel (a tile is 96 dependent symbol steps at the C2 geometry), so
// it is kept to ONE memory round trip: the constellation gather.  The two samples of the NEXT symbol are requested
// together with it, as a 3-sample window — the next symbol instant is n + ⌊mu + omega + mucorr⌋ with |mucorr| ≤ 0.1,
// i.e. one of two adjacent positions — and the soft symbol is stored fire-and-forget (the following wait is for the
// next iteration's loads, one full symbol step later).  No scratch; LDS only in the staged forms below.
// LDS staging of the tolerance tiles' samples (cu8 input only: LSDR_IN_CU8, nearest / linear sampler).  Every lane of a
// wavefront walks its own tile, so a per-symbol global load touches 64 different cache lines; with more than one wavefront
// per SIMD the lines of a CU's lanes no longer fit its L1 (nor, chip-wide, the L2s) and every 8-byte window costs a 128-byte
// line from memory (measured: 4 wavefronts per SIMD ran 6× slower per wavefront than one).  Instead the wavefront fetches,
// for all of its 64 tiles at once, the next kStage samples (+ look-ahead margin) with kStageLoads buffer→LDS loads of 1 KiB —
// every 128-byte line crosses once — and the per-symbol windows come out of LDS.  One row of kRowBytes per lane; the rows are
// filled in flat order (LDS-DMA writes lane·16 contiguous bytes), the global side of each 16-byte piece is per lane.  The
// buffer resource starts at the 16-byte boundary below the stream's first byte (a pipebuf<cu8> read pointer is only 2-byte
// aligned) and ends with the readable samples: pieces past the end come back as zeros and are never used.
// cf32 input stages the same way with shorter rows (16 samples + the interpolation partner: at 8 bytes per sample a per-symbol
// window was a 128-byte line per lane per FOUR symbols, half of them from memory, and the wavefront waited for the slowest lane on
// every step; the stage's last 16-byte piece is also the first piece of the NEXT stage's line, which so is on its way early).
template <int FMT> struct rx_stage;
template <> struct rx_stage<LSDR_IN_CU8> {
  static constexpr int kStage = 64;                                   // samples per stage (two stages per chunk)
  static constexpr int kStageMargin = 16;                             // look-ahead kept behind a stage (covers omega ≤ ≈ 10)
  static constexpr int kBps = 2;                                      // bytes per sample
  static constexpr int kRowBytes = kBps * (kStage + kStageMargin) + 16;   // + alignment slack
  static constexpr int kWindow = 8;                                   // bytes one window read takes out of a row
};
template <> struct rx_stage<LSDR_IN_CF32> {
  static constexpr int kStage = 16;
  static constexpr int kStageMargin = 1;                              // the interpolation partner of the stage's last sample
  static constexpr int kBps = 8;
  static constexpr int kRowBytes = kBps * (kStage + kStageMargin) + 8;    // (+ 8: the stream may start on an odd sample of a 16-byte piece)
  static constexpr int kWindow = 16;
};
static_assert(rx_stage<LSDR_IN_CU8>::kRowBytes % 16 == 0 && rx_stage<LSDR_IN_CF32>::kRowBytes % 16 == 0, "stage geometry");
static_assert(kChunk % rx_stage<LSDR_IN_CU8>::kStage == 0 && kChunk % rx_stage<LSDR_IN_CF32>::kStage == 0, "stage geometry");
typedef __attribute__((address_space(3))) void *rx_lds_ptr;

// ---- symbol timing of a wavefront's tiles, fed forward (tolerance tiles, omega ≥ 2) -------------------------------------------------
// A tile starts in the middle of the stream and has only its warm-up to find the symbol timing; at ≈ 4 samples per symbol a start half a
// symbol off sits on the Mueller & Müller detector's unstable point (sdr.h:822-840) and a few tiles in a thousand have not left it when
// their body begins: unreconciled seams, lost packets.  The envelope knows the timing without any loop: |x|² of a pulse-shaped stream has
// a line at the symbol rate whose phase is the symbol instants' position (Oerder & Meyr) — c = Σ w[n]·|x[n]|²·e^{−j2πn/omega} over kEstSpan
// samples (Hann-weighted: the window's own leakage of the DC term is what a rectangular one would add), instants at n ≡ τ (mod omega),
// τ = omega·arg(conj c)/2π.  One estimate at the wavefront's FIRST tile and one at its LAST, the 64 lanes sharing the samples of each
// (coalesced loads, a wave reduction); every lane takes its own timing from the two — the nominal omega carries the first estimate to its
// tile, the difference to the second one, spread evenly, is the clock error the nominal omega does not know.  Below 2 samples per symbol
// the line is aliased — and the detector needs no help there (its pull-in range covers the ±0.6 samples a tile can be off).
constexpr int kEstSpan = 512;
template <typename SRC>
__device__ __forceinline__ float rx_wave_timing(const SRC &src, long long first, long long end, float omega, int lane, float *quality) {
  float cr = 0.f, sr = 0.f, pw = 0.f;
  const float w = 1.0f / omega;
#pragma unroll
  for (int i = 0; i < kEstSpan / 64; ++i) {
    const int idx = i * 64 + lane;                                  // (lanes side by side: 512 B per load)
    if (first + idx < end) {
      const float2 x = src[first + idx];
      const float hann = 0.5f - 0.5f * __builtin_amdgcn_cosf((float)idx * (1.0f / kEstSpan));
      const float p = (x.x * x.x + x.y * x.y) * hann;
      const float rev = (float)idx * w;
      const float fr = rev - __builtin_floorf(rev);
      cr += p * __builtin_amdgcn_cosf(fr); sr += p * __builtin_amdgcn_sinf(fr); pw += p;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { cr += __shfl_xor(cr, d, 64); sr += __shfl_xor(sr, d, 64); pw += __shfl_xor(pw, d, 64); }
  *quality = pw > 0.f ? __builtin_sqrtf(cr * cr + sr * sr) / pw : 0.f;
  float rev = atan2f(sr, cr) * 0.15915494f;                          // power peaks where 2π·n/omega ≡ arg(Σ p·e^{+jθ})
  rev -= __builtin_floorf(rev);
  return rev * omega;                                                // first symbol instant at or behind `first`, in samples
}

template <int SAMP, bool ARITH, int FMT, bool LDS, bool HARD>
__device__ __forceinline__ void rx_tile_tol(const rx_tiled_args &a, unsigned j0, int lane, char *lds) {
  const bool valid = lane < (int)a.lanes_per_wave && j0 + (unsigned)lane < a.n_tiles;
  const rx_consts &C = a.C;
  const bool rrc_lds = SAMP == 2 && C.ncoeffs <= kRrcLdsTaps;
  // fed-forward symbol timing (all 64 lanes take part): per GROUP of 32 consecutive tiles — tiles 1 + 32g … 32g + 32, whatever the
  // lanes per wavefront, like the estimator maps: a run is the same bits with 32 or 64 tiles per wavefront — one estimate at the group's
  // first tile and one at its last
  float est_first = 0.f, est_step = 0.f;
  bool est_ok = false;
  unsigned est_j0 = j0;
  if (SAMP != 2 && C.omega >= 2.0f && (a.lanes_per_wave == 32u || a.lanes_per_wave == 64u)) {
    const long long end = (long long)a.total_chunks * kChunk + (SAMP == 1 ? 1 : 0);
    const typename in_stream<FMT>::type all = in_make<FMT>(a.in);
    for (unsigned g = 0; g < a.lanes_per_wave / 32u; ++g) {
      const unsigned jf = j0 + 32u * g;
      if (jf >= a.n_tiles) break;
      const unsigned jl = jf + 31u < a.n_tiles - 1 ? jf + 31u : a.n_tiles - 1;
      const long long wsf = ((long long)a.first_chunks + (long long)(jf - 1) * a.tile_chunks - a.warm_chunks) * kChunk;
      const long long wsl = wsf + (long long)(jl - jf) * a.tile_chunks * kChunk;
      float q0 = 0.f, q1 = 0.f, step = 0.f;
      const float t0 = rx_wave_timing(all, wsf, end, C.omega, lane, &q0);
      const bool ok = q0 > 0.02f && wsf + kEstSpan <= end;
      if (ok && jl > jf && wsl + kEstSpan <= end) {
        const float t1 = rx_wave_timing(all, wsl, end, C.omega, lane, &q1);
        if (q1 > 0.02f) {
          // what the nominal omega predicts at the last tile, against what is measured there: the difference (within half a symbol) is
          // the clock error over the group's span
          const double span = (double)(wsl - wsf), om = (double)C.omega;
          double pred = (double)t0 - span; pred -= om * __builtin_floor(pred / om);
          float d = t1 - (float)pred;
          d -= C.omega * __builtin_rintf(d / C.omega);
          step = d / (float)(jl - jf);
        }
      }
      if ((unsigned)lane / 32u == g || a.lanes_per_wave == 32u) { est_first = t0; est_step = step; est_ok = ok; est_j0 = jf; }
    }
  }
  if (!LDS && !valid) return;                            // (LDS: every lane takes part in the stage loads)
  const unsigned j = valid ? j0 + (unsigned)lane : j0;
  const unsigned long long c0 = a.first_chunks + (unsigned long long)(j - 1) * a.tile_chunks;
  unsigned long long c1 = c0 + a.tile_chunks;
  if (c1 > a.total_chunks) c1 = a.total_chunks;
  const unsigned long long cb = c0 - a.warm_chunks;
  const int nwarm = (int)a.warm_chunks, nchunks = valid ? (int)(c1 - cb) : 0;
  const int wave_chunks = LDS ? nwarm + (int)a.tile_chunks : nchunks;       // wave-uniform trip count of the chunk loop
  const int ra = SAMP == 1 ? 1 : (SAMP == 2 ? C.ncoeffs - 1 : 0);
  const int n_last = nchunks * kChunk - 1 + ra;          // last readable sample of this tile's span
  const typename in_stream<FMT>::type base = in_make<FMT>(a.in) + cb * kChunk;

  // LDS staging: this lane's row, the wave's buffer resource, the per-lane part of the source offsets
  typedef rx_stage<FMT> ST;
  constexpr int kStage = ST::kStage, kRowBytes = ST::kRowBytes, kStageLoads = ST::kRowBytes / 16;
  const char *row = nullptr;
  int delta = 0;
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned src_off[LDS ? kStageLoads : 1];
  if (LDS) {
    const unsigned long long addr = (unsigned long long)a.in;
    delta = (int)(addr & 15ull);
    const unsigned long long cb0 = a.first_chunks + (unsigned long long)(j0 - 1) * a.tile_chunks - a.warm_chunks;   // first tile of the wave
    // cu8: the resource spans the run (offsets clamp at 4 GiB); cf32: it starts at the wavefront's first tile, so that the
    // offsets stay small however long the run is
    const unsigned long long org = FMT == LSDR_IN_CU8 ? 0ull : cb0;
    // extent: up to the end of the 16-byte granule that holds the last readable sample (range checks are per dword, and a
    // granule never crosses a page), zeros beyond
    const unsigned long long bytes = (((a.total_chunks - org) * kChunk + (unsigned)ra) * (unsigned)ST::kBps + (unsigned)delta + 15ull) & ~15ull;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(reinterpret_cast<const char *>(a.in)) + org * (kChunk * ST::kBps) - delta, 0,
                                             (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
    row = lds + lane * kRowBytes;
#pragma unroll
    for (int q = 0; q < kStageLoads; ++q) {
      const unsigned f = (unsigned)q * 1024u + (unsigned)lane * 16u, r = f / (unsigned)kRowBytes, col = f - r * (unsigned)kRowBytes;
      const unsigned long long tile_byte = (cb0 - org + (unsigned long long)r * a.tile_chunks) * (kChunk * (unsigned long long)ST::kBps);
      // rows of tiles that do not exist, or offsets beyond 4 GiB, point past the end of the resource: zeros
      src_off[q] = (j0 + r < a.n_tiles && tile_byte + col < 0xfff00000ull) ? (unsigned)(tile_byte + col) : 0xfffffff0u;
    }
  }

  const rx_state_dev *S = a.state;                        // carried tracking state (wave-uniform)
  float freqw = S->freqw, agc = S->agc_gain, est_insp = S->est_insp, est_sp = S->est_sp, est_ep = S->est_ep;
  const float min_f = S->min_freqw, max_f = S->max_freqw;
  float fwin = 65536.0f / C.omega / 2048.0f;
  if (fwin < 8.f) fwin = 8.f;
  const float f_lo = freqw - fwin, f_hi = freqw + fwin;
  // Symbol timing at the tile's first sample, PREDICTED from the carried state: the run's first sample is S->mu samples in front of a symbol
  // instant and the instants are omega apart (the loop holds omega fixed, sdr.h:738-743), so the first instant at or behind sample ws is
  // omega − fmod(ws − S->mu, omega) away.  Exact for a stream at the nominal symbol rate — then a tile starts where the sequential loop is
  // and its warm-up has nothing to acquire; with a clock error e the prediction is off by e·ws samples, i.e. for long runs as good as the
  // mu = 0 every tile started with before (the warm-up acquires, as before).  Why: at ≈ 4 samples per symbol a tile that starts half a
  // symbol off sits on the timing detector's unstable point and may not have left it when its body begins — 0.2 % of the tiles of the
  // reference benchmark's 4.2-sps series (profiles/r06_sensitivity: unreconciled seams, lost packets); integer omega hid it.
  float mu = 0.f, phase = 0.f;
  {
    const double ws = (double)cb * kChunk - (double)S->mu, om = (double)C.omega;
    const double r = ws - om * __builtin_floor(ws / om);
    mu = (float)(r > 0.0 ? om - r : 0.0);
    if (est_ok) {
      // omega ≥ 2: the measured timing — unless the prediction agrees with it within a third of a sample: then the stream runs at the
      // nominal rate and the prediction is the better figure (it has no estimation noise; what matters is only never to start near the
      // detector's unstable point, half a symbol off)
      const double off = (double)(j - est_j0) * a.tile_chunks * kChunk;      // this tile's start behind its group's first
      double t = (double)est_first + (double)est_step * (double)(j - est_j0) - off;
      t -= om * __builtin_floor(t / om);
      float dm = (float)t - mu;
      dm -= C.omega * __builtin_rintf(dm / C.omega);
      if (__builtin_fabsf(dm) > 0.33f) mu = (float)t;
    }
    if (!(mu >= 0.f && mu < C.omega)) mu = 0.f;
  }
  float h0pr = 0.f, h0pi = 0.f, h0cr = 0.f, h0ci = 0.f, h1pr = 0.f, h1pi = 0.f, h1cr = 0.f, h1ci = 0.f;
  float h2pr = 0.f, h2pi = 0.f, h2cr = 0.f, h2ci = 0.f;
  const float kk = C.kest, k1 = 1 - C.kest;
  // (loop gains in registers: left in the argument record, a per-lane choice between two of them became a per-lane LOAD from it)
  const float acq_alpha = C.acq_alpha, freq_alpha = C.freq_alpha, acq_gain_mu = C.acq_gain_mu, gain_mu = C.gain_mu;
  const float freq_beta = C.freq_beta, omega = C.omega;
  const int acq_syms = C.acq_syms;
  rx_ema_map m; m.a = 1.f; m.bi = 0.f; m.bs = 0.f; m.be = 0.f;

  rx_tile_info ti;
  ti.has_pre = 0; ti.pre.cost = 0; ti.pre.symbol = 0; ti.pre.pad = 0; ti.n_warm = 0;
  ti.mu_begin = ti.phase_begin = 0.f;
  // a soft symbol is the low dword of its table entry {cost, symbol, 0}: one 4-byte store, always issued (a warm-up
  // symbol that nobody needs lands in the tile's seam row, which the last warm-up chunk rewrites) so that the number of
  // stores in flight is the same on every path and the wait for the next symbol's samples need not drain them
  unsigned last = 0;
  unsigned *const po = reinterpret_cast<unsigned *>(a.stage + (unsigned long long)j * a.stage_stride);
  unsigned *const pw = reinterpret_cast<unsigned *>(a.wstage + (unsigned long long)j * a.wstride);
  unsigned *const pw0 = reinterpret_cast<unsigned *>(a.wstage + (unsigned long long)j0 * a.wstride);
  unsigned cnt = 0, got = 0;
  // LSDR_SYM_HARD2: the decisions only, packed (rx_tiling.h): the word being filled, the last 16 symbols, the row, the snapshot
  // of the tail at the end of the warm-up
  unsigned hacc = 0, htail = 0, hwarm = 0, hnwarm = 0, hcnt = 0;
  // Soft symbols of the tile's BODY leave in groups of four: one 16-byte store per four symbol steps instead of four 4-byte ones (staged and
  // direct-load tiles alike: c3's direct-load tiles 576 → 609 GS/s).  A lane writes its own row of the staging buffer, so every store instruction of the wavefront touches 64 different cache lines whatever
  // its width — the REQUESTS are what the filter next door pays for (no symbol stores at all: filter launch 0.399 → 0.389 ms in the C2
  // pipeline, profiles/r06_bench/rx_ablation.txt), and this is a quarter of them.  (stage_stride is a multiple of four symbols: rows start on 16 bytes.)
  unsigned q0 = 0, q1 = 0, q2 = 0, q3 = 0, qn = 0;
  unsigned *bp = po;
  const bool quads = !HARD && !(a.dbg & 1u);
  unsigned *const hcol = HARD ? a.hstage + j : nullptr;      // this tile's column of the transposed staging

  int n = 0;                                              // current sample
  rx_window<FMT> win;
  if (!LDS) {
    win.load(base, 0, n_last);
    if (!HARD) *pw = 0u;   // (puts the loop entry in the same "three loads, then one store" state as the loop's back edge: see `*dp = raw.x`)
  } else {
    win.wn = -16;
  }
  for (int ci = 0; ci < wave_chunks; ++ci) {
    const bool active = !LDS || ci < nchunks;
    const bool body = ci >= nwarm, lastwarm = ci + 1 == nwarm;
    const int cend = (ci + 1) * kChunk;
    if (active && ci == nwarm) {
      ti.mu_begin = mu; ti.phase_begin = phase;
      ti.pre.cost = (int16_t)(last & 0xffffu); ti.pre.symbol = (uint8_t)(last >> 16); ti.has_pre = got ? 1u : 0u;
      hwarm = htail; hnwarm = got < 16u ? got : 16u;
    }
    const float samp_freqw = freqw;                       // sampler->update_freq(freqw), sdr.h:790
    // (the symbols of the earlier warm-up chunks are needed by nobody: every lane parks them on ONE word — the first one of the
    // wavefront's first seam row, which that tile's last warm-up chunk rewrites — so that their stores cost the texture-address unit
    // one cache line instead of 64)
    unsigned *dp = body ? po + cnt : (lastwarm ? pw : pw0);
    unsigned keep = (body || lastwarm) ? 1u : 0u;
    if (a.dbg & 1u) { dp = pw0; keep = 0u; }
    bool had = false;
    float2 sg = make_float2(0.f, 0.f), sv = make_float2(0.f, 0.f);
    int pt_re = 0, pt_im = 0;
    unsigned nsym = 0;
    auto skip = [&]() {
      if (!(mu < 1.f)) {
        const int rem = cend - n;
        int k = (int)mu;
        if (k < 1 || k > rem) k = rem;                    // (k < 1: NaN — the reference then just walks to the chunk end)
        mu -= (float)k; n += k; phase += (float)k * freqw;
      }
    };
    if (active) skip();
#pragma unroll 1
    for (int sb = 0; sb < (LDS ? kChunk / kStage : 1); ++sb) {
      const int s0 = ci * kChunk + sb * kStage;           // first sample of the stage (tile-relative)
      const int send = LDS ? s0 + kStage : cend;
      // window of sample i out of this lane's LDS row
      auto lds_window = [&](rx_window<FMT> &wd, int i) {
        int bo = ST::kBps * (i - s0) + delta;
        bo = bo < 0 ? 0 : (bo > kRowBytes - ST::kWindow ? kRowBytes - ST::kWindow : bo);
        rx_window_lds(wd, row, bo, delta, s0);
      };
      if (LDS) {
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(s0 * ST::kBps);   // (wave-uniform; says so to the compiler: no waterfall loop)
#pragma unroll
        for (int q = 0; q < kStageLoads; ++q)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (rx_lds_ptr)(size_t)(unsigned)(unsigned long long)(lds + q * 1024), 16, src_off[q], soff, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the stage has landed (single-wave workgroup: no barrier)
        asm volatile("" ::: "memory");
        // (cf32 rows keep no look-ahead: the window a lane carries over a stage boundary was read out of the OLD stage)
        if (active && n < send && (FMT == LSDR_IN_CF32 || !win.covers(n))) lds_window(win, n);
      }
      while (active && n < send) {
        if (SAMP != 2 && !win.covers(n) && !(a.dbg & 2u)) {   // window missed (rare: rounding at the ±0.1 edge): plain reload
          if (LDS) lds_window(win, n);
          else {
            win.load(base, n, n_last);
            __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0) HERE, so that the common path only waits for its window
          }
        }
        float2 p0, p1;
        win.get(n, p0, p1);
        const float2 sx0 = cmul(p0, ld_hwtrig::expi(a.T, trig_index(-phase)));
        if (SAMP == 2) {
          // fir_sampler::interp (sdr.h:646-665): polyphase branch (1−mu)·S of the matched filter over the next ⌈N/S⌉ samples;
          // plain loads (an ⌈N/S⌉-sample window does not pay), taps from the per-run table
          // Every lane walks its own tile: a vector-memory instruction of the wavefront touches 64 cache lines, and the tap loop made two of them per
          // tap (22 per symbol at 11 taps).  The taps now come out of LDS (the per-run table, copied in once per wavefront: rrc_lds) and the samples
          // two at a time (cf32: one 16-byte load per pair) — the same products added in the same order.
          float2 acc = make_float2(0.f, 0.f);
          const int SS = C.subsampling, N = C.ncoeffs;
          int px = n;
          int pc = (int)((1 - mu) * SS);
          auto tap = [&](int i) -> float2 { return rrc_lds ? reinterpret_cast<const float2 *>(lds)[i] : a.T.shifted_tol[i]; };
          if (FMT == LSDR_IN_CF32) {
            for (; pc + SS < N; pc += 2 * SS, px += 2) {
              const float4 xx = *reinterpret_cast<const float4 *>(reinterpret_cast<const float2 *>(base.p) + px);
              const float2 t0 = cmul(tap(pc), make_float2(xx.x, xx.y));
              acc.x += t0.x; acc.y += t0.y;
              const float2 t1 = cmul(tap(pc + SS), make_float2(xx.z, xx.w));
              acc.x += t1.x; acc.y += t1.y;
            }
          }
          for (; pc < N; pc += SS, ++px) {
            const float2 tt = cmul(tap(pc), base[px]);
            acc.x += tt.x; acc.y += tt.y;
          }
          sg = cmul(ld_hwtrig::expi(a.T, trig_index(-phase)), acc);
        } else if (SAMP == 1) {
          const float2 sx1 = cmul(p1, ld_hwtrig::expi(a.T, trig_index(-(phase + samp_freqw))));
          const float k0 = 1 - mu;
          sg = make_float2(sx0.x * k0 + sx1.x * mu, sx0.y * k0 + sx1.y * mu);
        } else {
          sg = sx0;
        }
        sv = make_float2(sg.x * agc, sg.y * agc);
        uint2 raw;
        if (!ARITH) raw = *reinterpret_cast<const uint2 *>(a.T.lut + lut_index(sv.x, sv.y));
        // window for the next symbol: ⌊mu + omega − 0.1⌋ samples ahead (at least one)
        int lo = (int)(mu + omega - 0.1f);
        lo = lo < 1 ? 1 : lo;
        rx_window<FMT> nxt;
        if (LDS) lds_window(nxt, n + lo);
        else if (a.dbg & 2u) { nxt = win; nxt.wn = n + lo; }
        else nxt.load(base, n + lo, n_last);
        lut_entry e;
        if (ARITH) {
          float hi = sv.x, hq = sv.y;
          lut_halve(hi, hq);
          const qpsk_decision qd = qpsk_decide((int)hi, (int)hq);
          e.cost = (int16_t)qd.cost; e.symbol = (uint8_t)qd.symbol; e.zero = 0;
          e.phase_error = (int16_t)qd.phase_error; e.pt_re = (int8_t)qd.pt_re; e.pt_im = (int8_t)qd.pt_im;
          raw.x = ((unsigned)qd.cost & 0xffffu) | (qd.symbol << 16);
        } else {
          e = lut_unpack(raw.x, raw.y);
        }
        if (HARD) {
          const unsigned hs = (raw.x >> 16) & 3u;
          htail = (htail << 2) | hs;
          if (body) {
            hacc = (hacc << 2) | hs;
            if ((++hcnt & 15u) == 0) hcol[(unsigned long long)((hcnt >> 4) - 1) * a.hpitch] = hacc;
          }
        } else if (quads && body) {
          q0 = q1; q1 = q2; q2 = q3; q3 = raw.x;
          if (++qn == 4u) {
            *reinterpret_cast<uint4 *>(bp) = make_uint4(q0, q1, q2, q3);
            bp += 4; qn = 0;
          }
        } else {
          *dp = raw.x; dp += keep;
        }
        last = raw.x;
        ++nsym;
        const bool acq = (int)(got + nsym) <= acq_syms && !body;
        phase += e.phase_error * (acq ? acq_alpha : freq_alpha);   // sdr.h:814-815
        freqw += e.phase_error * freq_beta;
        freqw = freqw < f_lo ? f_lo : freqw; freqw = freqw > f_hi ? f_hi : freqw;
        h2pr = h1pr; h2pi = h1pi; h2cr = h1cr; h2ci = h1ci;  // sdr.h:822-840
        h1pr = h0pr; h1pi = h0pi; h1cr = h0cr; h1ci = h0ci;
        h0pr = sv.x; h0pi = sv.y;
        pt_re = e.pt_re; pt_im = e.pt_im;
        had = true;
        h0cr = (float)pt_re; h0ci = (float)pt_im;
        const float muerr = ((h0pr - h2pr) * h1cr + (h0pi - h2pi) * h1ci) - ((h0cr - h2cr) * h1pr + (h0ci - h2ci) * h1pi);
        float mucorr = muerr * (acq ? acq_gain_mu : gain_mu);
        mucorr = mucorr < -0.1f ? -0.1f : mucorr; mucorr = mucorr > 0.1f ? 0.1f : mucorr;
        mu += mucorr;
        mu += omega;
        mu -= 1.f; phase += freqw; ++n;                   // the symbol's own sample step
        win = nxt;
        skip();
      }
    }
    if (active) {
      phase = fmod65536(phase);                           // sdr.h:855
      if (had) {
        const float insp = sg.x * sg.x + sg.y * sg.y;     // sdr.h:867-870
        est_insp = insp * kk + est_insp * k1;
        if (est_insp) agc = kCstlnAmp / __builtin_sqrtf(est_insp);
        const float evr = sv.x - pt_re, evi = sv.y - pt_im; // sdr.h:873-889
        float sig_power, ev_power;
        if (C.nsymbols == 2) {
          const float sig_real = (float)((double)(pt_re + pt_im) * 0.707);
          const float ev_real = (float)((double)(evr + evi) * 0.707);
          sig_power = sig_real * sig_real; ev_power = ev_real * ev_real;
        } else {
          sig_power = (float)(pt_re * pt_re + pt_im * pt_im);
          ev_power = evr * evr + evi * evi;
        }
        est_sp = sig_power * kk + est_sp * k1;
        est_ep = ev_power * kk + est_ep * k1;
        if (body) {
          m.a *= k1;
          m.bi = insp * kk + m.bi * k1; m.bs = sig_power * kk + m.bs * k1; m.be = ev_power * kk + m.be * k1;
        }
      }
      if (!C.allow_drift) {                               // sdr.h:895-898
        if (freqw < min_f || freqw > max_f) freqw = (max_f + min_f) / 2;
      }
      if (body) cnt += nsym; else got += nsym;
      if (lastwarm) ti.n_warm = nsym;
      if (body && a.meas) rx_tile_meas(a, cb + (unsigned long long)ci, j, freqw, m);
      if (body && a.cstln) a.cstln[cb + (unsigned long long)ci] = had ? sv : make_float2(__builtin_nanf(""), __builtin_nanf(""));   // sdr.h:861-864
    }
  }
  ti.mu_end = mu; ti.phase_end = phase; ti.count = cnt;
  if (quads && qn) {      // the last one to three symbols (what lies behind them in the 16 bytes is never read: ti.count says how many there are)
    const uint4 v = qn == 1u ? make_uint4(q3, 0u, 0u, 0u) : (qn == 2u ? make_uint4(q2, q3, 0u, 0u) : make_uint4(q1, q2, q3, 0u));
    *reinterpret_cast<uint4 *>(bp) = v;
  }
  if (valid) {
    if (HARD) {
      if (hcnt & 15u) hcol[(unsigned long long)(hcnt >> 4) * a.hpitch] = hacc << (2 * (16 - (hcnt & 15u)));
      rx_tile_info_h th;
      th.mu_begin = ti.mu_begin; th.phase_begin = ti.phase_begin; th.mu_end = mu; th.phase_end = phase;
      th.count = cnt; th.has_pre = ti.has_pre; th.n_warm = hnwarm; th.warm_tail = hwarm; th.body_tail = htail;
      a.hinfo[j] = th;
    } else {
      a.info[j] = ti;
    }
  }
  {
    // inclusive composition over the tiles of this wavefront in GROUPS of (at most) 32 (lane order = stream order; lanes without
    // a tile — the highest ones — hold the identity and are never read): lane L ends with maps[first of its group] … maps[L]
    // composed.  (Groups, not wavefronts: a run's estimators then come out bit for bit the same with 32 and with 64 tiles per
    // wavefront, i.e. whether or not the captures of a GPU share their launches.)
    rx_ema_map inc = m;
    const int gl = lane & 31;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      rx_ema_map o;
      o.a = __shfl_up(inc.a, d, 32); o.bi = __shfl_up(inc.bi, d, 32); o.bs = __shfl_up(inc.bs, d, 32); o.be = __shfl_up(inc.be, d, 32);
      if (gl >= d) inc = ema_then(o, inc);
    }
    rx_ema_map ex;
    ex.a = __shfl_up(inc.a, 1, 32); ex.bi = __shfl_up(inc.bi, 1, 32); ex.bs = __shfl_up(inc.bs, 1, 32); ex.be = __shfl_up(inc.be, 1, 32);
    if (gl == 0) { ex.a = 1.f; ex.bi = ex.bs = ex.be = 0.f; }
    if (valid) {
      a.ema[j] = ex;
      const unsigned grp = a.lanes_per_wave > 32u ? 1u + (blockIdx.x - 1u) * 2u + (unsigned)(lane >> 5) : blockIdx.x;
      if (gl == 31 || lane == (int)a.lanes_per_wave - 1 || j == a.n_tiles - 1) a.ema_wave[grp] = inc;
    }
  }
  if (valid && j == a.n_tiles - 1) {
    rx_state_dev *o = a.state_next;
    o->mu = mu; o->phase = phase; o->freqw = freqw; o->agc_gain = agc;
    o->est_insp = est_insp; o->est_sp = est_sp; o->est_ep = est_ep;
    o->min_freqw = min_f; o->max_freqw = max_f; o->samp_freqw = freqw; o->update_freq_phase = S->update_freq_phase;
    o->meas_count = (a.meas_base + a.total_chunks * kChunk) % C.meas_decimation;
    o->hist[0] = h0pr; o->hist[1] = h0pi; o->hist[2] = h0cr; o->hist[3] = h0ci;
    o->hist[4] = h1pr; o->hist[5] = h1pi; o->hist[6] = h1cr; o->hist[7] = h1ci;
    o->hist[8] = h2pr; o->hist[9] = h2pi; o->hist[10] = h2cr; o->hist[11] = h2ci;
  }
}

// One launch runs the tiles of up to kRxMulti independent captures (one receiver each, same configuration and run geometry:
// lsdr_rx_run_multi_async; an ordinary run is the case of one) — blockIdx.y picks the capture's argument record out of the
// kernel-argument segment.  (Four captures on four streams were five streams on the runtime's four hardware queues: two
// receivers shared a queue and their runs — four launches each per batch — serialised: that queue, 94 % busy, paced the C2
// pipeline, not the filter.)
constexpr int kRxMulti = 8;
struct rx_tiled_multi { rx_tiled_args a[kRxMulti]; };
template <int SAMP, bool ARITH, int FMT, bool LDS, bool HARD>
__global__ __launch_bounds__(64) void k_rx_tiles(rx_tiled_multi m) {
  __shared__ __attribute__((aligned(16))) char lds[LDS ? 64 * rx_stage<FMT>::kRowBytes : (SAMP == 2 ? kRrcLdsTaps * 8 : 16)];
  const rx_tiled_args &a = m.a[blockIdx.y];
  if (SAMP == 2 && a.C.ncoeffs <= kRrcLdsTaps) {      // fir_sampler: the run's shifted taps into LDS (rx_tile_tol reads them there)
    for (int i = (int)threadIdx.x; i < a.C.ncoeffs; i += 64) reinterpret_cast<float2 *>(lds)[i] = a.T.shifted_tol[i];
    __syncthreads();
  }
  // LSDR_RX_PRIO=1 (tuning hook): raised issue priority.  Next to fir_filter's streaming wavefronts the tiles then take 201 instead of
  // 284 us per C2 batch — and the filter's launch, which paces that pipeline, not a microsecond less; next to viterbi_sync (C3), which
  // paces THAT chain, they cost it 15 %.  Off.
  if (a.dbg & 4u) __builtin_amdgcn_s_setprio(3);
  if (blockIdx.x == 0) { if (threadIdx.x == 0) rx_tile_exact<SAMP, FMT, HARD>(a); }
  else rx_tile_tol<SAMP, ARITH, FMT, LDS, HARD>(a, 1u + (blockIdx.x - 1u) * a.lanes_per_wave, (int)threadIdx.x, lds);
}

// fir_sampler in the tiled mode: the tolerance tiles' shifted taps, rebuilt from the carried freqw before every run
// (do_update_freq, sdr.h:676-680: sc[i] = expi(−(freqw/S)(i − N/2))·c[i])
__global__ __launch_bounds__(256) void k_rx_fir_refresh(const rx_state_dev *state, const float2 *trig, const float *coeffs, int N, int S,
                                                        float2 *shifted_tol) {
  const float f = state->freqw / S;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float2 e = trig[trig_index(-f * (i - N / 2))];
    const float cc = coeffs[i];
    shifted_tol[i] = make_float2(e.x * cc, e.y * cc);
  }
}

// Scan of the wavefronts' estimator maps (one workgroup; a run has a few hundred wavefronts): final estimators / AGC of
// the run and the measurement slots.
//  * state ← state_next (end state of the last tile; tiles read `state` while they run, so it is only replaced here),
//    with est_insp/est_sp/est_ep = (all maps composed)(carried values) and agc_gain = 75/sqrt(est_insp) (sdr.h:870);
//  * slot q of `meas` holds the partial map of its tile up to the measurement instant: composed with the maps of the
//    preceding wavefronts and of the preceding tiles of its own wavefront (ex[]) it becomes the estimator values there.
constexpr unsigned kEmaThreads = 256;
// the tiles' estimator maps are composed in groups of min(lanes per wavefront, 32) tiles (rx_tile_tol); group 0 is the exact first tile
static unsigned rx_ema_group(unsigned lpw) { return lpw > 32u ? 32u : lpw; }
static unsigned rx_ema_groups(unsigned n_tiles, unsigned lpw) { return 1u + (n_tiles - 1u + rx_ema_group(lpw) - 1u) / rx_ema_group(lpw); }
__device__ __forceinline__ void rx_ema_body(const rx_ema_map *wave, unsigned n_waves, const rx_ema_map *ex, unsigned n_tiles,
                                            unsigned lanes_per_wave, const rx_state_dev *next, rx_state_dev *state,
                                            rx_meas *meas, unsigned nm) {
  __shared__ rx_ema_map s_pre[kEmaThreads];     // composition of everything before thread t's wavefronts
  __shared__ rx_ema_map s_wave[kEmaThreads / 64];
  const unsigned t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const unsigned per = (n_waves + kEmaThreads - 1) / kEmaThreads;
  const unsigned lo = t * per < n_waves ? t * per : n_waves, hi = lo + per < n_waves ? lo + per : n_waves;
  const float e_insp = state->est_insp, e_sp = state->est_sp, e_ep = state->est_ep;   // carried values
  rx_ema_map m; m.a = 1.f; m.bi = m.bs = m.be = 0.f;
  for (unsigned i = lo; i < hi; ++i) m = ema_then(m, wave[i]);
  rx_ema_map inc = m;                            // inclusive scan over the lanes of a wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    rx_ema_map o;
    o.a = __shfl_up(inc.a, d, 64); o.bi = __shfl_up(inc.bi, d, 64); o.bs = __shfl_up(inc.bs, d, 64); o.be = __shfl_up(inc.be, d, 64);
    if (lane >= (unsigned)d) inc = ema_then(o, inc);
  }
  if (lane == 63) s_wave[wv] = inc;
  __syncthreads();
  rx_ema_map before; before.a = 1.f; before.bi = before.bs = before.be = 0.f;
  for (unsigned i = 0; i < wv; ++i) before = ema_then(before, s_wave[i]);
  rx_ema_map exl;                                // exclusive: lanes before this one in the wave
  exl.a = __shfl_up(inc.a, 1, 64); exl.bi = __shfl_up(inc.bi, 1, 64); exl.bs = __shfl_up(inc.bs, 1, 64); exl.be = __shfl_up(inc.be, 1, 64);
  if (lane == 0) { exl.a = 1.f; exl.bi = exl.bs = exl.be = 0.f; }
  s_pre[t] = ema_then(before, exl);
  __syncthreads();
  if (t == kEmaThreads - 1) {
    const rx_ema_map all = ema_then(s_pre[t], m);
    rx_state_dev s = *next;
    s.est_insp = all.a * e_insp + all.bi; s.est_sp = all.a * e_sp + all.bs; s.est_ep = all.a * e_ep + all.be;
    if (s.est_insp) s.agc_gain = kCstlnAmp / __builtin_sqrtf(s.est_insp);
    *state = s;
  }
  for (unsigned q = t; q < nm; q += kEmaThreads) {
    rx_meas mm = meas[q];
    const unsigned j = mm.tile < n_tiles ? mm.tile : 0u;
    const unsigned w = j == 0 ? 0u : 1u + (j - 1u) / lanes_per_wave, owner = w / per;
    rx_ema_map pm = s_pre[owner];
    for (unsigned i = owner * per; i < w; ++i) pm = ema_then(pm, wave[i]);
    pm = ema_then(pm, ex[j]);
    rx_ema_map part; part.a = mm.a; part.bi = mm.est_insp; part.bs = mm.est_sp; part.be = mm.est_ep;
    const rx_ema_map f = ema_then(pm, part);
    mm.est_insp = f.a * e_insp + f.bi; mm.est_sp = f.a * e_sp + f.bs; mm.est_ep = f.a * e_ep + f.be;
    meas[q] = mm;
  }
}
__global__ __launch_bounds__(kEmaThreads) void k_rx_ema(const rx_ema_map *wave, unsigned n_waves, const rx_ema_map *ex, unsigned n_tiles,
                                                        unsigned lanes_per_wave, const rx_state_dev *next, rx_state_dev *state,
                                                        rx_meas *meas, unsigned nm) {
  rx_ema_body(wave, n_waves, ex, n_tiles, lanes_per_wave, next, state, meas, nm);
}
struct rx_ema_rec { const rx_ema_map *wave, *ex; const rx_state_dev *next; rx_state_dev *state; };
struct rx_ema_multi { rx_ema_rec c[kRxMulti]; };
__global__ __launch_bounds__(kEmaThreads) void k_rx_ema_multi(rx_ema_multi m, unsigned n_waves, unsigned n_tiles, unsigned lanes_per_wave) {
  const rx_ema_rec &c = m.c[blockIdx.x];
  rx_ema_body(c.wave, n_waves, c.ex, n_tiles, lanes_per_wave, c.next, c.state, nullptr, 0u);
}
// seam pass and compaction of several captures (rx_tiling.h bodies; blockIdx.y = capture)
struct rx_seam_rec {
  const rx_tile_info *info; rx_tile_fix *fix; rx_seam_part *part; const lsdr_softsymbol *stage, *wstage; lsdr_softsymbol *out;
  rx_state_dev *state; rx_seam_result *res;
};
struct rx_seam_multi { rx_seam_rec c[kRxMulti]; };
__global__ __launch_bounds__(kSeamBlock) void k_rx_seam_multi(rx_seam_multi m, unsigned n_tiles, float omega, int R, float quad,
                                                              unsigned stage_stride, unsigned wstride, const uint8_t *relabel) {
  const rx_seam_rec &c = m.c[blockIdx.y];
  rx_seam_body<rx_tile_info, lsdr_softsymbol>(c.info, c.fix, n_tiles, omega, R, quad, c.part, c.stage, stage_stride, c.wstage, wstride, relabel);
}
// Seam pass and estimator scan in ONE launch: the scan (one workgroup per capture) neither reads what the seam pass writes nor the
// other way round, and as a launch of its own it was 20 µs of a dependent chain next to fir_filter (4.8 µs alone).  blockIdx.x < the
// seam blocks: seam pass; the block behind them: rx_ema_body.  An ordinary run is the case of one capture (with its measurement slots).
static_assert(kEmaThreads == kSeamBlock, "one launch geometry for the seam pass and the estimator scan");
__global__ __launch_bounds__(kSeamBlock) void k_rx_seam_ema_multi(rx_seam_multi m, rx_ema_multi em, unsigned n_tiles, float omega, int R, float quad,
                                                                  unsigned stage_stride, unsigned wstride, const uint8_t *relabel,
                                                                  unsigned n_groups, unsigned ema_group, rx_meas *meas, unsigned nm) {
  const unsigned nsb = (n_tiles + kSeamBlock - 1) / kSeamBlock;
  if (blockIdx.x == nsb) {
    const rx_ema_rec &c = em.c[blockIdx.y];
    rx_ema_body(c.wave, n_groups, c.ex, n_tiles, ema_group, c.next, c.state, meas, nm);
    return;
  }
  const rx_seam_rec &c = m.c[blockIdx.y];
  rx_seam_body<rx_tile_info, lsdr_softsymbol>(c.info, c.fix, n_tiles, omega, R, quad, c.part, c.stage, stage_stride, c.wstage, wstride, relabel);
}
__global__ __launch_bounds__(64) void k_rx_compact_multi(rx_seam_multi m, unsigned n_tiles, int R, float quad, unsigned stage_stride,
                                                         const uint8_t *relabel) {
  const rx_seam_rec &c = m.c[blockIdx.y];
  rx_compact_body<lsdr_softsymbol, rx_state_dev>(c.stage, stage_stride, c.info, c.fix, c.part, relabel, n_tiles, R, quad, c.out, c.state, c.res);
}

// ---------------------------------------------------------------- exact receiver, one LANE per independent capture
// The recurrence cannot be parallelised inside a stream without giving up bit-exactness — across streams it can: lane l of
// a wavefront runs capture l with the reference's exact arithmetic (rx_chunk, the routine of the serial kernel; table
// look-ups as vector gathers from the same tables), 64 captures per wavefront, as many wavefronts as there are captures
// (BASELINE config 4's shape: many independent 2 MS/s captures).  Bit-exact soft symbols and loop state per capture.
struct rx_batch_args {
  const void *const *in;               // [n_streams] device pointers (cf32 or cu8 items, FMT)
  lsdr_softsymbol *const *out;         // [n_streams]
  unsigned long long chunks;           // the same number of 128-sample chunks for every capture
  rx_state_dev *states;                // [n_streams]
  unsigned long long *produced;        // [n_streams]
  unsigned n_streams;
  rx_consts C;
  rx_tables T;
};

template <int SAMP, int FMT>
__global__ __launch_bounds__(64) void k_rx_batch(rx_batch_args a) {
  const unsigned sidx = blockIdx.x * 64u + threadIdx.x;
  if (sidx >= a.n_streams) return;
  rx_state_dev s = a.states[sidx];
  const typename in_stream<FMT>::type pin = in_make<FMT>(a.in[sidx]);
  lsdr_softsymbol *po = a.out[sidx];
  unsigned long long nout = 0;
  // (a lane writes its own capture's symbols: 64 cache lines per store instruction — four symbols per 16-byte store where the capture's output
  // starts on 16 bytes, a quarter of the requests)
  const bool quads = ((size_t)po & 15u) == 0;
  unsigned q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  for (unsigned long long c = 0; c < a.chunks; ++c) {
    if (SAMP == 1) s.samp_freqw = s.freqw;     // sampler->update_freq(freqw), sdr.h:790
    bool wrote;
    unsigned cnt = 0;
    rx_chunk<SAMP, ld_vec>(a.T, a.C, s, pin + c * kChunk, [&](lsdr_softsymbol ss) {
      if (quads) {
        q0 = q1; q1 = q2; q2 = q3; q3 = *reinterpret_cast<const unsigned *>(&ss);
        if (((nout + cnt) & 3ull) == 3ull) *reinterpret_cast<uint4 *>(po + (nout + cnt - 3)) = make_uint4(q0, q1, q2, q3);
        ++cnt;
      } else {
        po[nout + cnt++] = ss;
      }
    }, nullptr, &wrote);
    nout += cnt;
    s.meas_count += kChunk;                    // sdr.h:905-913 (the measurement pipes are not wired in the batch form)
    while (s.meas_count >= a.C.meas_decimation) s.meas_count -= a.C.meas_decimation;
  }
  if (quads && (nout & 3ull)) {      // the last one to three symbols
    unsigned *pu = reinterpret_cast<unsigned *>(po);
    const unsigned r = (unsigned)(nout & 3ull);
    if (r == 1u) pu[nout - 1] = q3;
    else if (r == 2u) { pu[nout - 2] = q2; pu[nout - 1] = q3; }
    else { pu[nout - 3] = q1; pu[nout - 2] = q2; pu[nout - 1] = q3; }
  }
  a.states[sidx] = s;
  a.produced[sidx] = nout;
}

#include "notch_detect.h"
#include "rxb_device.h"

}  // namespace

struct lsdr_rx {
  lsdr_ctx *ctx;
  lsdr_rx_cfg cfg;
  std::vector<float> coeffs;
  lsdr::cstln_tables tabs;
  // receiver parameters kept on the host exactly like the reference's members
  float omega, min_omega, max_omega;
  rx_state_dev st;          // host mirror of the device state
  rx_state_dev st_initial;  // the state lsdr_rx_create left (lsdr_rx_reset)
  bool st_dirty_host;       // host copy newer than device
  // device
  float2 *d_trig;
  lut_entry *d_lut;
  float *d_coeffs;
  float2 *d_shifted, *d_shifted_tol;
  rx_state_dev *d_state;
  unsigned long long *d_counters;
  rx_meas *d_meas; size_t meas_cap;
  float2 *d_cstln; size_t cstln_cap;
  // tiled mode
  lsdr_softsymbol *d_stage; size_t stage_cap;
  lsdr_softsymbol *d_wstage; size_t wstage_cap;
  unsigned *d_hstage; size_t hstage_cap;      // LSDR_SYM_HARD2: packed rows, [tiles_cap] info records
  rx_tile_info_h *d_hinfo;
  size_t out_sym_offset;                     // LSDR_SYM_HARD2: where the next run starts writing in `out` (symbols)
  // measurement hook (lsdr_rx_tile_time): HIP events around the k_rx_tiles launch of every queued run while enabled
  bool time_on; hipEvent_t tev0[8], tev1[8]; bool tev_set[8]; double time_ms; unsigned time_n;
  rx_tile_info *d_info; rx_tile_fix *d_fix; size_t tiles_cap;
  rx_ema_map *d_ema;              // [tiles_cap] per-tile exclusive prefix inside its wavefront
  rx_ema_map *d_ema_wave;         // [tiles_cap + 1] per-wavefront estimator maps
  rx_state_dev *d_state_next;     // end state of a tiled run before k_rx_ema installs it
  rx_state_dev *h_snap;           // pinned: lsdr_rx_snapshot_async targets (kSnapSlots of them)
  static constexpr unsigned kSnapSlots = 4;
  uint8_t *d_relabel;
  struct rx_seam_result *d_seam;
  struct rx_seam_part *d_part;
  std::vector<uint8_t> relabel;   // [nrotations][256]
  unsigned last_tiles, last_dup, last_miss, last_badseam;  // diagnostics of the last tiled run
  // queued (asynchronous) tiled runs: results land in a pinned ring, one event per slot
  static const int kRing = 8;
  struct rx_seam_result *h_res;        // pinned [kRing]
  struct rx_seam_result *h_res_dev;    // the same memory as seen by the device
  hipEvent_t ev[kRing];
  unsigned ring_tiles[kRing];
  int ring_head, ring_count;           // oldest outstanding slot, number outstanding
  float retired_freq_tap;              // freq_tap after the most recently retired queued run
  bool st_stale_host;                  // device state newer than the host mirror `st`
  bool qpsk_arith;                     // tolerance tiles decide by arithmetic (QPSK, verified against the table at create time)
  unsigned arith_max_dpe;              // largest |phase_error − table| seen by that verification
};

static void rx_state_export(const rx_state_dev &s, lsdr_rx_state *st) {
  st->mu = s.mu; st->phase = s.phase; st->freqw = s.freqw; st->agc_gain = s.agc_gain;
  st->est_insp = s.est_insp; st->est_sp = s.est_sp; st->est_ep = s.est_ep;
  st->freq_tap = s.freqw / 65536;  // refresh_freq_tap, sdr.h:919-921
  st->min_freqw = s.min_freqw; st->max_freqw = s.max_freqw;
  st->meas_count = s.meas_count;
  memcpy(st->hist, s.hist, sizeof(st->hist));
}

// sdr.h:755-770
static void rx_update_freq_limits(lsdr_rx *r, bool have_cstln) {
  int n = 4;
  if (have_cstln) {
    switch (r->tabs.nsymbols) {
      case 2: n = 2; break;
      case 4: n = 4; break;
      case 8: n = 8; break;
      case 16: n = 12; break;
      case 32: n = 16; break;
      default: n = 4; break;
    }
  }
  r->st.min_freqw = r->st.freqw - 65536 / r->max_omega / n / 2;
  r->st.max_freqw = r->st.freqw + 65536 / r->max_omega / n / 2;
}
// sdr.h:738-743 (tol = 10e-6 as a float parameter)
static void rx_set_omega(lsdr_rx *r, float omega, bool have_cstln) {
  float tol = (float)10e-6;
  r->omega = omega;
  r->min_omega = omega * (1 - tol);
  r->max_omega = omega * (1 + tol);
  rx_update_freq_limits(r, have_cstln);
}
// sdr.h:745-749
static void rx_set_freq(lsdr_rx *r, float freq, bool have_cstln) {
  r->st.freqw = freq * 65536;
  rx_update_freq_limits(r, have_cstln);
}

static int rx_push_state(lsdr_rx *r) {
  if (!r->st_dirty_host) return LSDR_OK;
  LSDR_HIP(hipMemcpyAsync(r->d_state, &r->st, sizeof(rx_state_dev), hipMemcpyHostToDevice, r->ctx->stream));
  LSDR_HIP(hipStreamSynchronize(r->ctx->stream));
  r->st_dirty_host = false;
  return LSDR_OK;
}

static void rx_fill_consts(const lsdr_rx *r, rx_consts &C, rx_tables &T) {
  C.omega = r->omega;
  C.freq_alpha = (float)0.04;                                                    // sdr.h:776
  C.freq_beta = (float)(0.0012 / (double)r->omega * (double)r->cfg.pll_adjustment);  // sdr.h:777
  C.gain_mu = (float)(0.02 / (double)(kCstlnAmp * kCstlnAmp) * 2);               // sdr.h:778
  C.kest = r->cfg.kest;
  C.allow_drift = r->cfg.allow_drift;
  C.nsymbols = r->tabs.nsymbols;
  C.meas_decimation = r->cfg.meas_decimation;
  C.ncoeffs = r->cfg.ncoeffs;
  C.subsampling = r->cfg.subsampling;
  {
    // Acquisition gear of the tolerance tiles' warm-up (tuning hooks; defaults chosen from tools/rx_tol_report.py sweeps)
    const char *e;
    C.acq_syms = (e = getenv("LSDR_RX_ACQ_SYMS")) ? atoi(e) : 0;
    C.acq_alpha = C.freq_alpha * ((e = getenv("LSDR_RX_ACQ_ALPHA")) ? (float)atof(e) : 1.f);
    C.acq_gain_mu = C.gain_mu * ((e = getenv("LSDR_RX_ACQ_MU")) ? (float)atof(e) : 1.f);
  }
  T.trig = r->d_trig; T.lut = r->d_lut; T.coeffs = r->d_coeffs; T.shifted = r->d_shifted; T.shifted_tol = r->d_shifted_tol;
}

static int rx_pull_state(lsdr_rx *r) {   // refresh the host mirror after queued runs
  if (!r->st_stale_host) return LSDR_OK;
  LSDR_HIP(hipMemcpyAsync(&r->st, r->d_state, sizeof(rx_state_dev), hipMemcpyDeviceToHost, r->ctx->stream));
  LSDR_HIP(hipStreamSynchronize(r->ctx->stream));
  r->st_stale_host = false;
  return LSDR_OK;
}

// LSDR_RX_TILED: see the file header.  Not bit-exact: every tile but the first
// re-acquires timing/phase during its warm-up; seams are reconciled on the device.
// rx_tiled_enqueue puts one run on the stream (tiles → seam → compaction → results into a pinned ring
// slot) without waiting; rx_tiled_wait retires the oldest queued run.
// One queued run in three steps, so that several receivers can share the launches (lsdr_rx_run_multi_async):
//   rx_tiled_plan    sizes, scratch, the argument record (no launch); chunks == 0: nothing to do
//   rx_tiled_launch  the run's kernels on the receiver's stream (or: the multi-capture launches over several plans)
//   rx_tiled_commit  ring slot, completion event, host-side bookkeeping
struct rx_plan {
  rx_tiled_args a;
  lsdr_softsymbol *out;
  size_t chunks, nm;
  unsigned n_tiles, blocks, lpw, stage_stride, sym_per_chunk;
  unsigned long long hpitch, meas_base, md;
  bool want_meas, want_cstln, use_lds, hard;
  int slot;
};

static int rx_tiled_plan(lsdr_rx *r, unsigned share, const void *in, size_t n_in, lsdr_softsymbol *out, size_t cap_out,
                         bool want_meas, size_t meas_cap, size_t cstln_cap, rx_plan *P) {
  lsdr_ctx *c = r->ctx;
  LSDR_HIP(hipSetDevice(c->device));
  P->out = out; P->chunks = 0; P->nm = 0; P->n_tiles = 0;
  if (r->ring_count == lsdr_rx::kRing) { lsdr_set_error("cstln_receiver: too many queued runs (lsdr_rx_wait first)"); return LSDR_E_ARG; }
  const int ra = lsdr_rx_readahead(r);
  // Defaults: a warm-up of ≈ 64 symbols (whole chunks) — the TS-level yield is flat from 32 to 256 symbols of
  // warm-up down to loss of lock (profiles/r02_sensitivity/) — and tiles twice as long as the warm-up.
  unsigned Wc = r->cfg.tile_warmup ? r->cfg.tile_warmup / kChunk : (unsigned)((64.0f * r->omega + kChunk - 1) / kChunk);
  if (!r->cfg.tile_warmup && Wc < 1) Wc = 1;
  unsigned Lc = r->cfg.tile_len ? r->cfg.tile_len / kChunk : 2 * Wc;
  LSDR_ARG(Lc >= 1 && Wc >= 1);
  // Symbols per chunk are bounded by 128/(omega - max_mucorr) + 1 (mu advances by at
  // least omega-0.1 per symbol, sdr.h:834-840).
  const unsigned sym_per_chunk = (unsigned)(kChunk / (r->omega - 0.1f)) + 2;
  size_t chunks = n_in >= (size_t)(kChunk + ra) ? (n_in - ra) / kChunk : 0;
  // (+1 symbol per seam: a repaired seam may re-insert a warm-up symbol)
  if ((size_t)(sym_per_chunk + 1) * chunks > cap_out) chunks = cap_out / (sym_per_chunk + 1);
  const bool want_cstln = cstln_cap > 0;
  if (want_cstln && chunks > cstln_cap) chunks = cstln_cap;     // one sampled point per chunk at most (sdr.h:785-788 gate)
  {
    // LDS-staged cu8 tiles address their samples through a buffer resource with 32-bit byte offsets (rows beyond 0xfff00000 are
    // pointed past the end and read zeros): a run is cut to what those offsets reach — `consumed` reports the partial run and the
    // caller comes back with the rest, as with any other limit.  (LSDR_RX_LDS_SPAN: test hook, bytes.)
    static const bool no_lds0 = getenv("LSDR_RX_NO_LDS") != nullptr;
    static const unsigned long long span = getenv("LSDR_RX_LDS_SPAN") ? strtoull(getenv("LSDR_RX_LDS_SPAN"), nullptr, 0) : 0xfff00000ull;
    const bool lds_tiles = r->cfg.in_format == LSDR_IN_CU8 && r->cfg.sampler != LSDR_SAMP_FIR && !no_lds0 && r->omega <= 8.f;
    if (lds_tiles) {
      const unsigned long long max_samples = (span - 64) / 2;
      if ((unsigned long long)chunks * kChunk + (unsigned)ra > max_samples) chunks = (size_t)((max_samples - (unsigned)ra) / kChunk);
    }
  }
  const int slot = (r->ring_head + r->ring_count) % lsdr_rx::kRing;
  P->slot = slot; P->chunks = chunks; P->want_meas = want_meas; P->want_cstln = want_cstln;
  if (!chunks) return LSDR_OK;   // nothing to do: the commit still occupies a slot so that wait() pairs with run_async()
  // tile 0 (one lane, the exact recurrence) only has to reach the point where tile 1's warm-up can start
  const unsigned first = Wc;
  unsigned n_tiles = 1;
  if (chunks > first) n_tiles += (unsigned)((chunks - first + Lc - 1) / Lc);
  const unsigned stage_stride = ((first > Lc ? first : Lc) * sym_per_chunk + 3u + 4u) & ~3u;      // (whole 16-byte groups, one spare: rx_tile_tol's last store)

  const bool hard = r->cfg.out_format == LSDR_SYM_HARD2;
  const unsigned hstride = stage_stride / 16 + 2;          // words per tile
  const unsigned long long hpitch = ((unsigned long long)n_tiles + 63) & ~63ull;   // transposed staging: word w of tile j at [w·hpitch + j]
  if (r->tiles_cap < n_tiles || (!hard && (r->stage_cap < (size_t)n_tiles * stage_stride || r->wstage_cap < (size_t)n_tiles * sym_per_chunk)) ||
      (hard && r->hstage_cap < (size_t)hpitch * hstride)) {
    // scratch grows: queued runs may still be using the old buffers
    LSDR_HIP(hipStreamSynchronize(c->stream));
  }
  if (r->tiles_cap < n_tiles) {
    (void)hipFree(r->d_info); (void)hipFree(r->d_fix); (void)hipFree(r->d_part); (void)hipFree(r->d_ema);
    LSDR_HIP(hipMalloc((void **)&r->d_ema, n_tiles * sizeof(rx_ema_map)));
    (void)hipFree(r->d_ema_wave);
    LSDR_HIP(hipMalloc((void **)&r->d_ema_wave, ((size_t)n_tiles + 1) * sizeof(rx_ema_map)));
    LSDR_HIP(hipMalloc((void **)&r->d_part, ((n_tiles + kSeamBlock - 1) / kSeamBlock) * sizeof(rx_seam_part)));
    LSDR_HIP(hipMalloc((void **)&r->d_info, n_tiles * sizeof(rx_tile_info)));
    LSDR_HIP(hipMalloc((void **)&r->d_fix, n_tiles * sizeof(rx_tile_fix)));
    (void)hipFree(r->d_hinfo); r->d_hinfo = nullptr;
    if (hard) LSDR_HIP(hipMalloc((void **)&r->d_hinfo, n_tiles * sizeof(rx_tile_info_h)));
    r->tiles_cap = n_tiles;
  }
  if (hard && r->hstage_cap < (size_t)hpitch * hstride) {
    (void)hipFree(r->d_hstage);
    LSDR_HIP(hipMalloc((void **)&r->d_hstage, (size_t)hpitch * hstride * sizeof(unsigned)));
    r->hstage_cap = (size_t)hpitch * hstride;
  }
  if (!hard && r->stage_cap < (size_t)n_tiles * stage_stride) {
    (void)hipFree(r->d_stage);
    LSDR_HIP(hipMalloc((void **)&r->d_stage, (size_t)n_tiles * stage_stride * sizeof(lsdr_softsymbol)));
    r->stage_cap = (size_t)n_tiles * stage_stride;
  }
  if (!hard && r->wstage_cap < (size_t)n_tiles * sym_per_chunk) {
    (void)hipFree(r->d_wstage);
    LSDR_HIP(hipMalloc((void **)&r->d_wstage, (size_t)n_tiles * sym_per_chunk * sizeof(lsdr_softsymbol)));
    r->wstage_cap = (size_t)n_tiles * sym_per_chunk;
  }
  const unsigned long long md = r->cfg.meas_decimation;
  const unsigned long long meas_base = r->st.meas_count;   // kept current on the host even while `st` is stale
  const size_t nm = (size_t)((meas_base + chunks * kChunk) / md - meas_base / md);
  P->nm = nm; P->meas_base = meas_base; P->md = md;
  if (want_meas && nm > meas_cap) { lsdr_set_error("cstln_receiver(tiled): measurement buffers too small"); return LSDR_E_ARG; }
  if (want_meas && r->meas_cap < nm + 1) {
    LSDR_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(r->d_meas);
    LSDR_HIP(hipMalloc((void **)&r->d_meas, (nm + 1) * sizeof(rx_meas)));
    r->meas_cap = nm + 1;
  }
  if (want_cstln && r->cstln_cap < chunks) {
    LSDR_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(r->d_cstln);
    LSDR_HIP(hipMalloc((void **)&r->d_cstln, chunks * sizeof(float2)));
    r->cstln_cap = chunks;
  }
  int rc = rx_push_state(r);
  if (rc) return rc;

  rx_tiled_args &a = P->a;
  a.in = in;
  a.total_chunks = chunks;
  a.first_chunks = first; a.tile_chunks = Lc; a.warm_chunks = Wc;
  a.n_tiles = n_tiles;
  a.stage_stride = stage_stride;
  a.stage = r->d_stage;
  a.wstage = r->d_wstage; a.wstride = sym_per_chunk;
  a.hstage = r->d_hstage; a.hpitch = hpitch; a.hinfo = r->d_hinfo;
  a.info = r->d_info;
  a.ema = r->d_ema;
  a.ema_wave = r->d_ema_wave;
  a.state = r->d_state;
  a.state_next = r->d_state_next;
  a.meas = want_meas ? r->d_meas : nullptr;
  a.meas_base = meas_base;
  a.cstln = want_cstln ? r->d_cstln : nullptr;
  rx_fill_consts(r, a.C, a.T);
  // 32 tiles per wavefront: few enough wavefronts that the whole batch is resident in one round even while fir_filter's
  // persistent workgroups hold most of the register file (C2 bench, streams overlapped: 8 → 292, 16 → 298, 32 → 310,
  // 64 → 302 GS/s whole-job).  With more than ≈ 12 K tiles in a launch (short tiles, or the captures of a GPU sharing it: `share`)
  // 64 per wavefront keeps the wavefront count where fir_filter is disturbed least (128-sample tiles, 17.5 K of them:
  // 32 → fir 0.160 ms per launch, 64 → 0.149 ms, same whole-job rate).
  int lpw = (unsigned long long)n_tiles * share > 12288ull ? 64 : 32;
  {
    static const char *const e = getenv("LSDR_RX_LANES");   // tuning hook: tiles per wavefront
    if (e) lpw = atoi(e);
    if (lpw < 1 || (lpw > 32 && lpw != 64)) lpw = (unsigned long long)n_tiles * share > 12288ull ? 64 : 32;   // (estimator groups: ≤ 32, or two of 32)
    a.lanes_per_wave = (unsigned)lpw;
    static const char *const d = LSDR_MEASURE_ENV("LSDR_RX_DBG");   // measure build only: timing-only tiles
    static const bool prio = getenv("LSDR_RX_PRIO") && atoi(getenv("LSDR_RX_PRIO"));
    a.dbg = (d ? (unsigned)atoi(d) & 3u : 0u) | (prio ? 4u : 0u);
  }
  // nearest / linear sampler: the tiles' samples are staged through LDS (see rx_tile_tol) — cu8 always; cf32 when all 64 rows of a
  // stage belong to tiles of the wavefront and the launch is small enough (≤ 1024 wavefronts) that its 9 KiB per wavefront do not
  // decide how many are resident: C3's 4 Gi-sample batches are 8.7 K wavefronts, and next to fir_filter's 114 KiB per CU only
  // four of them fit a CU — direct loads (no LDS) ran that chain 12 % faster.  LSDR_RX_NO_LDS=1 keeps the direct loads (A/B
  // measurements), =2 for cf32 input only
  static const int no_lds = getenv("LSDR_RX_NO_LDS") ? atoi(getenv("LSDR_RX_NO_LDS")) : 0;
  const bool lds_fmt = r->cfg.in_format == LSDR_IN_CU8 ? (no_lds != 1 && r->omega <= 8.f)
                                                       : (no_lds == 0 && lpw == 64 && (unsigned long long)n_tiles * share <= 65536ull &&
                                                          ((unsigned long long)in & 7ull) == 0);
  const bool use_lds = r->cfg.sampler != LSDR_SAMP_FIR && lds_fmt;
  const unsigned blocks = 1 + (n_tiles - 1 + (unsigned)lpw - 1) / (unsigned)lpw;
  P->n_tiles = n_tiles; P->blocks = blocks; P->lpw = (unsigned)lpw; P->stage_stride = stage_stride; P->sym_per_chunk = sym_per_chunk;
  P->hpitch = hpitch; P->hard = hard;
  if (hard && !(use_lds && r->cfg.in_format == LSDR_IN_CU8)) { lsdr_set_error("cstln_receiver: LSDR_SYM_HARD2 needs cu8 input with the nearest or linear sampler"); return LSDR_E_UNSUPPORTED; }
  // k_rx_compact_h lets a partial output word be finished by the NEXT tile from the previous tile's column only: every tile must
  // hold at least two words' worth of symbols (32) after a dropped first one
  if (hard && (float)Lc * kChunk / (r->omega + 0.1f) < 34.f) {
    lsdr_set_error("cstln_receiver: LSDR_SYM_HARD2 needs tiles of at least 34 symbols (tile_len %u samples at omega %.2f)", Lc * kChunk, (double)r->omega);
    return LSDR_E_ARG;
  }
  P->use_lds = use_lds;
  return LSDR_OK;
}

static int rx_tiled_launch(lsdr_rx *r, const rx_plan &P) {
  lsdr_ctx *c = r->ctx;
  const rx_tiled_args &a = P.a;
  const unsigned blocks = P.blocks, n_tiles = P.n_tiles, stage_stride = P.stage_stride, sym_per_chunk = P.sym_per_chunk;
  const unsigned long long hpitch = P.hpitch;
  const bool hard = P.hard, use_lds = P.use_lds, want_meas = P.want_meas;
  const size_t nm = P.nm;
  const int lpw = (int)P.lpw, slot = P.slot;
  lsdr_softsymbol *const out = P.out;
  rx_tiled_multi tm1;
  for (int i = 0; i < kRxMulti; ++i) tm1.a[i] = a;
#define LSDR_RX_LAUNCH_F(S, A, F, L, H) hipLaunchKernelGGL((k_rx_tiles<S, A, F, L, H>), dim3(blocks), dim3(64), 0, c->stream, tm1)
#define LSDR_RX_LAUNCH(S, A) do { if (hard) LSDR_RX_LAUNCH_F(S, A, LSDR_IN_CU8, (S != 2), (S != 2)); \
                                  else if (use_lds && r->cfg.in_format == LSDR_IN_CU8) LSDR_RX_LAUNCH_F(S, A, LSDR_IN_CU8, (S != 2), false); \
                                  else if (use_lds) LSDR_RX_LAUNCH_F(S, A, LSDR_IN_CF32, (S != 2), false); \
                                  else if (r->cfg.in_format == LSDR_IN_CU8) LSDR_RX_LAUNCH_F(S, A, LSDR_IN_CU8, false, false); \
                                  else LSDR_RX_LAUNCH_F(S, A, LSDR_IN_CF32, false, false); } while (0)
#define LSDR_RX_LAUNCH_S(S) do { if (r->qpsk_arith) LSDR_RX_LAUNCH(S, true); else LSDR_RX_LAUNCH(S, false); } while (0)
  if (r->cfg.sampler == LSDR_SAMP_FIR)
    hipLaunchKernelGGL(k_rx_fir_refresh, dim3(1), dim3(256), 0, c->stream, (const rx_state_dev *)r->d_state, (const float2 *)r->d_trig,
                       (const float *)r->d_coeffs, r->cfg.ncoeffs, r->cfg.subsampling, r->d_shifted_tol);
  if (r->time_on) { LSDR_HIP(hipEventRecord(r->tev0[slot], c->stream)); }
  if (r->cfg.sampler == LSDR_SAMP_NEAREST) LSDR_RX_LAUNCH_S(0);
  else if (r->cfg.sampler == LSDR_SAMP_LINEAR) LSDR_RX_LAUNCH_S(1);
  else LSDR_RX_LAUNCH_S(2);
  r->tev_set[slot] = r->time_on;
  if (r->time_on) { LSDR_HIP(hipEventRecord(r->tev1[slot], c->stream)); }
#undef LSDR_RX_LAUNCH_S
#undef LSDR_RX_LAUNCH
#undef LSDR_RX_LAUNCH_F
  LSDR_HIP(hipGetLastError());
  // ---- estimators (AGC, MER) of the run — scan of the tiles' maps, installs the end state —, seam pass, compaction: all on the stream
  const int R = r->tabs.nrotations;
  const float quad = 65536.0f / R;
  if (hard) {
    hipLaunchKernelGGL(k_rx_ema, dim3(1), dim3(kEmaThreads), 0, c->stream, (const rx_ema_map *)r->d_ema_wave, rx_ema_groups(n_tiles, (unsigned)lpw),
                       (const rx_ema_map *)r->d_ema, n_tiles, rx_ema_group((unsigned)lpw), (const rx_state_dev *)r->d_state_next, r->d_state,
                       want_meas ? r->d_meas : nullptr, want_meas ? (unsigned)nm : 0u);
    hipLaunchKernelGGL(k_rx_seam_h, dim3((n_tiles + kSeamBlock - 1) / kSeamBlock), dim3(kSeamBlock), 0, c->stream,
                       (const rx_tile_info_h *)r->d_hinfo, r->d_fix, n_tiles, r->omega, R, quad, r->d_part, (const uint8_t *)r->d_relabel);
    hipLaunchKernelGGL((k_rx_compact_h<rx_state_dev>), dim3((n_tiles + 63) / 64), dim3(64), 0, c->stream, (const unsigned *)r->d_hstage, hpitch,
                       (const rx_tile_info_h *)r->d_hinfo, (const rx_tile_fix *)r->d_fix, (const rx_seam_part *)r->d_part,
                       (const uint8_t *)r->d_relabel, n_tiles, R, quad, reinterpret_cast<unsigned *>(out),
                       (unsigned long long)r->out_sym_offset, r->d_state, r->h_res_dev + slot);
  } else {
    rx_seam_multi sm;
    rx_ema_multi em;
    for (int i = 0; i < kRxMulti; ++i) {
      sm.c[i] = rx_seam_rec{r->d_info, r->d_fix, r->d_part, r->d_stage, r->d_wstage, out, r->d_state, r->h_res_dev + slot};   // totals go straight into the pinned ring slot
      em.c[i] = rx_ema_rec{r->d_ema_wave, r->d_ema, r->d_state_next, r->d_state};
    }
    hipLaunchKernelGGL(k_rx_seam_ema_multi, dim3((n_tiles + kSeamBlock - 1) / kSeamBlock + 1, 1), dim3(kSeamBlock), 0, c->stream, sm, em, n_tiles,
                       r->omega, R, quad, stage_stride, sym_per_chunk, (const uint8_t *)r->d_relabel, rx_ema_groups(n_tiles, (unsigned)lpw),
                       rx_ema_group((unsigned)lpw), want_meas ? r->d_meas : nullptr, want_meas ? (unsigned)nm : 0u);
    hipLaunchKernelGGL(k_rx_compact_multi, dim3((n_tiles + kCompactTiles - 1) / kCompactTiles, 1), dim3(64), 0, c->stream, sm, n_tiles, R, quad,
                       stage_stride, (const uint8_t *)r->d_relabel);
  }
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

static int rx_tiled_commit(lsdr_rx *r, const rx_plan &P, hipStream_t stream, size_t *consumed) {
  const int slot = P.slot;
  if (!P.chunks) {
    r->h_res[slot].total = 0; r->h_res[slot].rot_final = 0; r->h_res[slot].ndup = r->h_res[slot].nmiss = r->h_res[slot].nbad = 0;
    r->h_res[slot].freq_tap = r->retired_freq_tap;
    r->ring_tiles[slot] = 0; r->tev_set[slot] = false;
    LSDR_HIP(hipEventRecord(r->ev[slot], stream));
    ++r->ring_count;
    *consumed = 0;
    return LSDR_OK;
  }
  LSDR_HIP(hipEventRecord(r->ev[slot], stream));
  r->ring_tiles[slot] = P.n_tiles;
  ++r->ring_count;
  r->st_stale_host = true;
  r->st.meas_count = (P.meas_base + P.chunks * kChunk) % P.md;   // what the last tile writes on the device
  *consumed = P.chunks * kChunk;
  return LSDR_OK;
}

static int rx_tiled_enqueue(lsdr_rx *r, const void *in, size_t n_in, lsdr_softsymbol *out, size_t cap_out,
                            size_t *consumed, bool want_meas, size_t meas_cap, size_t *nm_out, size_t cstln_cap = 0,
                            size_t *chunks_out = nullptr) {
  *consumed = 0;
  if (nm_out) *nm_out = 0;
  rx_plan P;
  LSDR_TRY(rx_tiled_plan(r, 1u, in, n_in, out, cap_out, want_meas, meas_cap, cstln_cap, &P));
  if (chunks_out) *chunks_out = P.chunks;
  if (P.chunks) LSDR_TRY(rx_tiled_launch(r, P));
  LSDR_TRY(rx_tiled_commit(r, P, r->ctx->stream, consumed));
  if (nm_out) *nm_out = P.nm;
  return LSDR_OK;
}

// Several receivers, one set of launches: see k_rx_tiles_multi.  The receivers must live on ONE context (stream), be configured
// alike (soft symbols out, same sampler / input format / tile geometry / constellation) and get equally long inputs; anything
// else is queued receiver by receiver — the same results either way.  consumed[i] = what receiver i takes (they can differ on
// the receiver-by-receiver paths: another omega with cap_out binding, the cu8 LDS-span cut).  EVERY run is planned before
// anything is launched: an argument error leaves no receiver queued.
static int rx_tiled_enqueue_multi(lsdr_rx *const *rs, unsigned n, const void *const *ins, size_t n_in, lsdr_softsymbol *const *outs,
                                  size_t cap_out, size_t *consumed) {
  for (unsigned i = 0; i < n; ++i) consumed[i] = 0;
  bool alike = n >= 2 && n <= (unsigned)kRxMulti;
  for (unsigned i = 1; i < n && alike; ++i) {
    const lsdr_rx *a = rs[0], *b = rs[i];
    alike = a->ctx == b->ctx && a->cfg.sampler == b->cfg.sampler && a->cfg.in_format == b->cfg.in_format &&
            a->cfg.out_format == b->cfg.out_format && a->cfg.tile_len == b->cfg.tile_len && a->cfg.tile_warmup == b->cfg.tile_warmup &&
            a->cfg.cstln == b->cfg.cstln && a->omega == b->omega && a->qpsk_arith == b->qpsk_arith && a->cfg.ncoeffs == b->cfg.ncoeffs &&
            a->cfg.subsampling == b->cfg.subsampling && a->tabs.nrotations == b->tabs.nrotations;
  }
  if (alike) alike = rs[0]->cfg.out_format != LSDR_SYM_HARD2 && rs[0]->cfg.sampler != LSDR_SAMP_FIR && !rs[0]->time_on;
  std::vector<rx_plan> Pv(n);
  rx_plan *P = Pv.data();
  for (unsigned i = 0; i < n; ++i) LSDR_TRY(rx_tiled_plan(rs[i], alike ? n : 1u, ins[i], n_in, outs[i], cap_out, false, 0, 0, &P[i]));
  for (unsigned i = 1; i < n && alike; ++i)
    alike = P[i].chunks == P[0].chunks && P[i].n_tiles == P[0].n_tiles && P[i].blocks == P[0].blocks && P[i].lpw == P[0].lpw &&
            P[i].stage_stride == P[0].stage_stride && P[i].use_lds == P[0].use_lds;
  if (!alike || !P[0].chunks) {      // receiver by receiver, from the plans
    for (unsigned i = 0; i < n; ++i) {
      if (P[i].chunks) LSDR_TRY(rx_tiled_launch(rs[i], P[i]));
      LSDR_TRY(rx_tiled_commit(rs[i], P[i], rs[i]->ctx->stream, &consumed[i]));
    }
    return LSDR_OK;
  }
  lsdr_rx *r = rs[0];
  lsdr_ctx *c = r->ctx;
  static const bool skip = LSDR_MEASURE_ENV("LSDR_RX_SKIP") != nullptr;     // measure build only: no receiver kernels at all (results are garbage)
  if (skip) {
    for (unsigned i = 0; i < n; ++i) LSDR_TRY(rx_tiled_commit(rs[i], P[i], c->stream, &consumed[i]));
    return LSDR_OK;
  }
  rx_tiled_multi tm;
  rx_ema_multi em;
  rx_seam_multi sm;
  for (unsigned i = 0; i < n; ++i) {
    tm.a[i] = P[i].a;
    em.c[i] = rx_ema_rec{rs[i]->d_ema_wave, rs[i]->d_ema, rs[i]->d_state_next, rs[i]->d_state};
    sm.c[i] = rx_seam_rec{rs[i]->d_info, rs[i]->d_fix, rs[i]->d_part, rs[i]->d_stage, rs[i]->d_wstage, outs[i], rs[i]->d_state,
                          rs[i]->h_res_dev + P[i].slot};
  }
  for (unsigned i = n; i < (unsigned)kRxMulti; ++i) { tm.a[i] = P[0].a; em.c[i] = em.c[0]; sm.c[i] = sm.c[0]; }
  const unsigned blocks = P[0].blocks, n_tiles = P[0].n_tiles;
  const bool use_lds = P[0].use_lds;
#define LSDR_RXM_LAUNCH_F(S, A, F, L) hipLaunchKernelGGL((k_rx_tiles<S, A, F, L, false>), dim3(blocks, n), dim3(64), 0, c->stream, tm)
#define LSDR_RXM_LAUNCH(S, A) do { if (use_lds && r->cfg.in_format == LSDR_IN_CU8) LSDR_RXM_LAUNCH_F(S, A, LSDR_IN_CU8, true); \
                                   else if (use_lds) LSDR_RXM_LAUNCH_F(S, A, LSDR_IN_CF32, true); \
                                   else if (r->cfg.in_format == LSDR_IN_CU8) LSDR_RXM_LAUNCH_F(S, A, LSDR_IN_CU8, false); \
                                   else LSDR_RXM_LAUNCH_F(S, A, LSDR_IN_CF32, false); } while (0)
#define LSDR_RXM_LAUNCH_S(S) do { if (r->qpsk_arith) LSDR_RXM_LAUNCH(S, true); else LSDR_RXM_LAUNCH(S, false); } while (0)
  if (r->cfg.sampler == LSDR_SAMP_NEAREST) LSDR_RXM_LAUNCH_S(0);
  else LSDR_RXM_LAUNCH_S(1);
#undef LSDR_RXM_LAUNCH_S
#undef LSDR_RXM_LAUNCH
#undef LSDR_RXM_LAUNCH_F
  LSDR_HIP(hipGetLastError());
  const int R = r->tabs.nrotations;
  const float quad = 65536.0f / R;
  hipLaunchKernelGGL(k_rx_seam_ema_multi, dim3((n_tiles + kSeamBlock - 1) / kSeamBlock + 1, n), dim3(kSeamBlock), 0, c->stream, sm, em, n_tiles, r->omega,
                     R, quad, P[0].stage_stride, P[0].sym_per_chunk, (const uint8_t *)r->d_relabel, rx_ema_groups(n_tiles, P[0].lpw),
                     rx_ema_group(P[0].lpw), (rx_meas *)nullptr, 0u);
  hipLaunchKernelGGL(k_rx_compact_multi, dim3((n_tiles + kCompactTiles - 1) / kCompactTiles, n), dim3(64), 0, c->stream, sm, n_tiles, R, quad,
                     P[0].stage_stride, (const uint8_t *)r->d_relabel);
  LSDR_HIP(hipGetLastError());
  for (unsigned i = 0; i < n; ++i) LSDR_TRY(rx_tiled_commit(rs[i], P[i], c->stream, &consumed[i]));
  return LSDR_OK;
}

static int rx_tiled_wait(lsdr_rx *r, size_t *produced) {
  if (!r->ring_count) { lsdr_set_error("cstln_receiver: no queued run"); return LSDR_E_ARG; }
  const int slot = r->ring_head;
  LSDR_HIP(hipEventSynchronize(r->ev[slot]));
  if (r->tev_set[slot]) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r->tev0[slot], r->tev1[slot]) == hipSuccess) { r->time_ms += ms; ++r->time_n; }
    r->tev_set[slot] = false;
  }
  const rx_seam_result sr = r->h_res[slot];
  r->ring_head = (r->ring_head + 1) % lsdr_rx::kRing;
  --r->ring_count;
  r->last_tiles = r->ring_tiles[slot]; r->last_dup = sr.ndup; r->last_miss = sr.nmiss; r->last_badseam = sr.nbad;
  r->retired_freq_tap = sr.freq_tap;
  *produced = (size_t)sr.total;
  return LSDR_OK;
}

static int rx_run_tiled(lsdr_rx *r, const void *in, size_t n_in, lsdr_softsymbol *out, size_t cap_out,
                        size_t *consumed, size_t *produced, float *freq_out, float *ss_out, float *mer_out,
                        size_t meas_cap, size_t *n_meas, lsdr_cf32 *cstln_out, size_t cstln_cap, size_t *n_cstln) {
  if (r->ring_count) { lsdr_set_error("cstln_receiver: queued runs outstanding (lsdr_rx_wait first)"); return LSDR_E_ARG; }
  const bool want_meas = freq_out || ss_out || mer_out;
  size_t nm = 0, chunks = 0;
  int rc = rx_tiled_enqueue(r, in, n_in, out, cap_out, consumed, want_meas, meas_cap, &nm, cstln_out ? cstln_cap : 0, &chunks);
  if (rc) return rc;
  rc = rx_tiled_wait(r, produced);
  if (rc) return rc;
  rc = rx_pull_state(r);
  if (rc) return rc;
  if (want_meas && nm) {
    std::vector<rx_meas> m(nm);
    LSDR_HIP(hipMemcpy(m.data(), r->d_meas, nm * sizeof(rx_meas), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nm; ++i) {
      if (freq_out) freq_out[i] = m[i].freqw / 65536;
      if (ss_out) ss_out[i] = sqrtf(m[i].est_insp);
      if (mer_out) mer_out[i] = m[i].est_ep ? 10 * logf(m[i].est_sp / m[i].est_ep) / logf(10) : 0;
    }
  }
  if (n_meas) *n_meas = want_meas ? nm : 0;
  if (cstln_out && cstln_cap && chunks) {     // one point per chunk that produced a symbol, in stream order
    std::vector<float2> pts(chunks);
    LSDR_HIP(hipMemcpy(pts.data(), r->d_cstln, chunks * sizeof(float2), hipMemcpyDeviceToHost));
    size_t k = 0;
    for (size_t i = 0; i < chunks && k < cstln_cap; ++i)
      if (pts[i].x == pts[i].x) { cstln_out[k].re = pts[i].x; cstln_out[k].im = pts[i].y; ++k; }
    if (n_cstln) *n_cstln = k;
  }
  return LSDR_OK;
}

extern "C" {

#ifdef LSDR_RX_TRACE
int lsdr_rx_probe_read(unsigned long long *host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rx_probe), sizeof(g_rx_probe)) == hipSuccess ? 0 : -1;
}
int lsdr_rx_probe_reset(void) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_rx_probe), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

int lsdr_rx_create(lsdr_ctx *c, const lsdr_rx_cfg *cfg, lsdr_rx **out) {
  LSDR_ARG(c && cfg && out);
  LSDR_ARG(cfg->sampler >= LSDR_SAMP_NEAREST && cfg->sampler <= LSDR_SAMP_FIR);
  LSDR_ARG(cfg->sampler != LSDR_SAMP_FIR || (cfg->ncoeffs > 0 && cfg->coeffs_host && cfg->subsampling >= 1));
  LSDR_ARG(cfg->omega > 0 && cfg->meas_decimation >= 1);
  LSDR_ARG(cfg->mode == LSDR_RX_SERIAL || cfg->mode == LSDR_RX_TILED);
  LSDR_ARG(cfg->in_format == LSDR_IN_CF32 || cfg->in_format == LSDR_IN_CU8);
  LSDR_ARG(cfg->out_format == LSDR_SYM_SOFT || cfg->out_format == LSDR_SYM_HARD2);
  if (cfg->out_format == LSDR_SYM_HARD2 && (cfg->mode != LSDR_RX_TILED || cfg->cstln != LSDR_QPSK || cfg->in_format != LSDR_IN_CU8 ||
                                            cfg->sampler == LSDR_SAMP_FIR)) {
    lsdr_set_error("cstln_receiver: LSDR_SYM_HARD2 is the tiled QPSK receiver on cu8 input (nearest / linear sampler)");
    return LSDR_E_UNSUPPORTED;
  }
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_rx *r = new lsdr_rx();
  r->ctx = c;
  r->cfg = *cfg;
  if (cfg->sampler == LSDR_SAMP_FIR) {
    r->coeffs.assign(cfg->coeffs_host, cfg->coeffs_host + cfg->ncoeffs);
    r->cfg.coeffs_host = r->coeffs.data();
  }
  int ns = lsdr::build_cstln(cfg->cstln, cfg->fec, r->tabs);
  if (ns < 0) {
    delete r;
    lsdr_set_error("cstln_receiver: unsupported constellation/code rate %d/%d", cfg->cstln, cfg->fec);
    return LSDR_E_ARG;
  }
  if (cfg->harden)  // cstln_lut::harden, sdr.h:564-571
    for (auto &v : r->tabs.cost) v = v < 0 ? -1 : (v > 0 ? 1 : 0);

  // Constructor sequence of the reference (sdr.h:709-736 then leandvb.cc:476-487):
  // ctor: est_insp=amp², agc_gain=1, mu=phase=0, set_omega(1), set_freq(0) with cstln==NULL;
  // then cstln is assigned, set_omega(Fs/Fm), optional set_freq(Ftune/Fs).
  memset(&r->st, 0, sizeof(r->st));
  r->st.est_insp = kCstlnAmp * kCstlnAmp;
  r->st.agc_gain = 1;
  rx_set_omega(r, 1, false);
  rx_set_freq(r, 0, false);
  rx_set_omega(r, cfg->omega, true);
  if (cfg->freq) rx_set_freq(r, cfg->freq, true);
  r->st_dirty_host = true;
  r->st_initial = r->st;

  // Tolerance tiles may decide QPSK symbols by arithmetic instead of the table gather — only if the arithmetic IS the
  // table: symbol, cost and constellation point identical for all 65536 entries, phase_error within ±2 table units.
  r->qpsk_arith = false; r->arith_max_dpe = 0;
  if (cfg->cstln == LSDR_QPSK && !cfg->harden && !getenv("LSDR_RX_NO_ARITH")) {
    bool ok = r->tabs.nsymbols == 4;
    unsigned worst = 0;
    for (int i = 0; ok && i < 65536; ++i) {
      const int I = (int8_t)(i >> 8), Q = (int8_t)(i & 255);
      const qpsk_decision d = qpsk_decide(I, Q);
      ok = d.cost == r->tabs.cost[i] && d.symbol == r->tabs.symbol[i] && d.pt_re == r->tabs.symbols[d.symbol][0] &&
           d.pt_im == r->tabs.symbols[d.symbol][1];
      int dpe = (int)(int16_t)d.phase_error - (int)r->tabs.phase_error[i];
      if (dpe < 0) dpe = -dpe;
      if ((unsigned)dpe > worst) worst = (unsigned)dpe;
    }
    r->qpsk_arith = ok && worst <= 2;
    r->arith_max_dpe = worst;
  }

  // Tables → HBM.
  std::vector<lsdr_cf32> trig(65536);
  lsdr_trig16_table(trig.data());
  std::vector<lut_entry> lut(65536);
  for (int i = 0; i < 65536; ++i) {
    lut_entry e;
    e.cost = r->tabs.cost[i];
    e.symbol = r->tabs.symbol[i];
    e.zero = 0;
    e.phase_error = r->tabs.phase_error[i];
    e.pt_re = r->tabs.symbols[e.symbol][0];
    e.pt_im = r->tabs.symbols[e.symbol][1];
    lut[i] = e;
  }
  LSDR_HIP(hipMalloc((void **)&r->d_trig, 65536 * sizeof(float2)));
  LSDR_HIP(hipMalloc((void **)&r->d_lut, 65536 * sizeof(lut_entry)));
  LSDR_HIP(hipMemcpy(r->d_trig, trig.data(), 65536 * sizeof(float2), hipMemcpyHostToDevice));
  LSDR_HIP(hipMemcpy(r->d_lut, lut.data(), 65536 * sizeof(lut_entry), hipMemcpyHostToDevice));
  r->d_coeffs = nullptr;
  r->d_shifted = nullptr; r->d_shifted_tol = nullptr;
  if (cfg->sampler == LSDR_SAMP_FIR) {
    LSDR_HIP(hipMalloc((void **)&r->d_coeffs, cfg->ncoeffs * sizeof(float)));
    LSDR_HIP(hipMalloc((void **)&r->d_shifted, cfg->ncoeffs * sizeof(float2)));
    LSDR_HIP(hipMemcpy(r->d_coeffs, r->coeffs.data(), cfg->ncoeffs * sizeof(float), hipMemcpyHostToDevice));
    LSDR_HIP(hipMemset(r->d_shifted, 0, cfg->ncoeffs * sizeof(float2)));
    LSDR_HIP(hipMalloc((void **)&r->d_shifted_tol, cfg->ncoeffs * sizeof(float2)));
    LSDR_HIP(hipMemset(r->d_shifted_tol, 0, cfg->ncoeffs * sizeof(float2)));
  }
  LSDR_HIP(hipMalloc((void **)&r->d_state, sizeof(rx_state_dev)));
  LSDR_HIP(hipMalloc((void **)&r->d_counters, 8 * sizeof(unsigned long long)));
  r->d_meas = nullptr; r->meas_cap = 0;
  r->d_cstln = nullptr; r->cstln_cap = 0;
  r->d_stage = nullptr; r->stage_cap = 0;
  r->d_info = nullptr; r->d_fix = nullptr; r->d_part = nullptr; r->tiles_cap = 0;
  r->d_ema = nullptr; r->d_ema_wave = nullptr; r->h_snap = nullptr;
  LSDR_HIP(hipMalloc((void **)&r->d_state_next, sizeof(rx_state_dev)));
  LSDR_HIP(hipHostMalloc((void **)&r->h_snap, lsdr_rx::kSnapSlots * sizeof(rx_state_dev), hipHostMallocDefault));
  r->d_wstage = nullptr; r->wstage_cap = 0;
  r->d_hstage = nullptr; r->hstage_cap = 0; r->d_hinfo = nullptr; r->out_sym_offset = 0;
  r->time_on = false; r->time_ms = 0; r->time_n = 0;
  for (int i = 0; i < lsdr_rx::kRing; ++i) { r->tev0[i] = r->tev1[i] = nullptr; r->tev_set[i] = false; }
  r->retired_freq_tap = r->st.freqw / 65536;
  r->h_res = nullptr; r->ring_head = 0; r->ring_count = 0; r->st_stale_host = false;
  for (int i = 0; i < lsdr_rx::kRing; ++i) r->ev[i] = nullptr;
  r->last_tiles = r->last_dup = r->last_miss = r->last_badseam = 0;
  // Relabel tables for the tiled mode: relabel[k][s] = symbol whose constellation point is
  // point[s] rotated by +k·(360°/nrotations) (nearest point; exact for the PSK/APSK/QAM sets).
  {
    const int R = r->tabs.nrotations, ns = r->tabs.nsymbols;
    r->relabel.assign((size_t)R * 256, 0);
    for (int k = 0; k < R; ++k) {
      double ang = 2 * M_PI * k / R, ca = cos(ang), sa = sin(ang);
      for (int s = 0; s < ns; ++s) {
        double x = r->tabs.symbols[s][0] * ca - r->tabs.symbols[s][1] * sa;
        double y = r->tabs.symbols[s][0] * sa + r->tabs.symbols[s][1] * ca;
        int best = 0; double bd = 1e30;
        for (int t = 0; t < ns; ++t) {
          double dx = x - r->tabs.symbols[t][0], dy = y - r->tabs.symbols[t][1];
          double d = dx * dx + dy * dy;
          if (d < bd) { bd = d; best = t; }
        }
        r->relabel[(size_t)k * 256 + s] = (uint8_t)best;
      }
    }
    LSDR_HIP(hipMalloc((void **)&r->d_seam, sizeof(rx_seam_result)));
    LSDR_HIP(hipHostMalloc((void **)&r->h_res, lsdr_rx::kRing * sizeof(rx_seam_result), hipHostMallocDefault));
    LSDR_HIP(hipHostGetDevicePointer((void **)&r->h_res_dev, r->h_res, 0));
    for (int i = 0; i < lsdr_rx::kRing; ++i) LSDR_HIP(hipEventCreateWithFlags(&r->ev[i], hipEventDisableTiming));
    LSDR_HIP(hipMalloc((void **)&r->d_relabel, r->relabel.size()));
    LSDR_HIP(hipMemcpy(r->d_relabel, r->relabel.data(), r->relabel.size(), hipMemcpyHostToDevice));
  }
  *out = r;
  return LSDR_OK;
}

void lsdr_rx_destroy(lsdr_rx *r) {
  if (!r) return;
  (void)hipStreamSynchronize(r->ctx->stream);
  (void)hipFree(r->d_trig); (void)hipFree(r->d_lut);
  (void)hipFree(r->d_coeffs); (void)hipFree(r->d_shifted); (void)hipFree(r->d_shifted_tol);
  (void)hipFree(r->d_state); (void)hipFree(r->d_counters);
  (void)hipFree(r->d_meas); (void)hipFree(r->d_cstln);
  (void)hipFree(r->d_stage); (void)hipFree(r->d_info); (void)hipFree(r->d_fix); (void)hipFree(r->d_relabel); (void)hipFree(r->d_seam); (void)hipFree(r->d_part); (void)hipFree(r->d_wstage);
  (void)hipFree(r->d_hstage); (void)hipFree(r->d_hinfo);
  (void)hipFree(r->d_ema); (void)hipFree(r->d_ema_wave); (void)hipFree(r->d_state_next);
  if (r->h_snap) (void)hipHostFree(r->h_snap);
  if (r->h_res) (void)hipHostFree(r->h_res);
  for (int i = 0; i < lsdr_rx::kRing; ++i) if (r->ev[i]) (void)hipEventDestroy(r->ev[i]);
  for (int i = 0; i < lsdr_rx::kRing; ++i) { if (r->tev0[i]) (void)hipEventDestroy(r->tev0[i]); if (r->tev1[i]) (void)hipEventDestroy(r->tev1[i]); }
  delete r;
}

int lsdr_rx_readahead(const lsdr_rx *r) {
  if (!r) return 0;
  switch (r->cfg.sampler) {
    case LSDR_SAMP_NEAREST: return 0;
    case LSDR_SAMP_LINEAR: return 1;
    default: return r->cfg.ncoeffs - 1;
  }
}

int lsdr_rx_get_state(lsdr_rx *r, lsdr_rx_state *st) {
  LSDR_ARG(r && st);
  { int rc = rx_pull_state(r); if (rc) return rc; }
  rx_state_export(r->st, st);  // host mirror is refreshed after every run
  return LSDR_OK;
}

int lsdr_rx_decision_mode(const lsdr_rx *r, int *arithmetic, unsigned *max_phase_error_delta) {
  LSDR_ARG(r);
  if (arithmetic) *arithmetic = r->qpsk_arith ? 1 : 0;
  if (max_phase_error_delta) *max_phase_error_delta = r->arith_max_dpe;
  return LSDR_OK;
}

int lsdr_rx_snapshot_async_slot(lsdr_rx *r, unsigned slot) {
  LSDR_ARG(r && slot < lsdr_rx::kSnapSlots);
  LSDR_HIP(hipSetDevice(r->ctx->device));
  if (r->st_dirty_host) { int rc = rx_push_state(r); if (rc) return rc; }
  LSDR_HIP(hipMemcpyAsync(r->h_snap + slot, r->d_state, sizeof(rx_state_dev), hipMemcpyDeviceToHost, r->ctx->stream));
  return LSDR_OK;
}
int lsdr_rx_snapshot_async(lsdr_rx *r) { return lsdr_rx_snapshot_async_slot(r, 0); }

int lsdr_rx_get_snapshot_slot(lsdr_rx *r, unsigned slot, lsdr_rx_state *st) {
  LSDR_ARG(r && st && slot < lsdr_rx::kSnapSlots);
  LSDR_HIP(hipStreamSynchronize(r->ctx->stream));
  rx_state_export(r->h_snap[slot], st);
  return LSDR_OK;
}
int lsdr_rx_get_snapshot(lsdr_rx *r, lsdr_rx_state *st) { return lsdr_rx_get_snapshot_slot(r, 0, st); }

int lsdr_rx_tiled_stats(const lsdr_rx *r, unsigned *tiles, unsigned *dup, unsigned *miss, unsigned *bad_seams) {
  LSDR_ARG(r);
  if (tiles) *tiles = r->last_tiles;
  if (dup) *dup = r->last_dup;
  if (miss) *miss = r->last_miss;
  if (bad_seams) *bad_seams = r->last_badseam;
  return LSDR_OK;
}

int lsdr_rx_reset(lsdr_rx *r) {   // the loop state right after lsdr_rx_create (a new capture begins); no host wait
  LSDR_ARG(r);
  if (r->ring_count) { lsdr_set_error("cstln_receiver: queued runs outstanding (lsdr_rx_wait first)"); return LSDR_E_ARG; }
  r->st = r->st_initial;
  r->st_stale_host = false;
  r->st_dirty_host = false;
  r->retired_freq_tap = r->st_initial.freqw / 65536;      // what refresh_freq_tap (sdr.h:919-921) shows for the new capture, not the old one's estimate
  r->last_tiles = r->last_dup = r->last_miss = r->last_badseam = 0;
  return lsdr_stage_h2d(r->ctx, r->d_state, &r->st, sizeof(rx_state_dev));
}

int lsdr_rx_set_state(lsdr_rx *r, const lsdr_rx_state *st) {
  LSDR_ARG(r && st);
  if (r->ring_count) { lsdr_set_error("cstln_receiver: queued runs outstanding (lsdr_rx_wait first)"); return LSDR_E_ARG; }
  { int rc = rx_pull_state(r); if (rc) return rc; }
  rx_state_dev &s = r->st;
  s.mu = st->mu; s.phase = st->phase; s.freqw = st->freqw; s.agc_gain = st->agc_gain;
  s.est_insp = st->est_insp; s.est_sp = st->est_sp; s.est_ep = st->est_ep;
  s.min_freqw = st->min_freqw; s.max_freqw = st->max_freqw;
  s.meas_count = st->meas_count;
  memcpy(s.hist, st->hist, sizeof(s.hist));
  r->st_dirty_host = true;
  return LSDR_OK;
}

int lsdr_rx_run_async(lsdr_rx *r, const void *in, size_t n_in, lsdr_softsymbol *out, size_t cap_out, size_t *consumed) {
  LSDR_ARG(r && consumed && (in || !n_in) && out);
  if (r->cfg.mode != LSDR_RX_TILED) { lsdr_set_error("cstln_receiver: lsdr_rx_run_async needs LSDR_RX_TILED"); return LSDR_E_UNSUPPORTED; }
  return rx_tiled_enqueue(r, in, n_in, out, cap_out, consumed, false, 0, nullptr);
}

int lsdr_rx_run_async_hs2(lsdr_rx *r, const void *in, size_t n_in, uint32_t *out_words, size_t out_sym_offset, size_t cap_out,
                          size_t *consumed) {
  LSDR_ARG(r && consumed && (in || !n_in) && out_words);
  if (r->cfg.mode != LSDR_RX_TILED || r->cfg.out_format != LSDR_SYM_HARD2) {
    lsdr_set_error("cstln_receiver: lsdr_rx_run_async_hs2 needs LSDR_RX_TILED with LSDR_SYM_HARD2");
    return LSDR_E_UNSUPPORTED;
  }
  r->out_sym_offset = out_sym_offset;
  const int rc = rx_tiled_enqueue(r, in, n_in, reinterpret_cast<lsdr_softsymbol *>(out_words), cap_out, consumed, false, 0, nullptr);
  r->out_sym_offset = 0;
  return rc;
}

int lsdr_rx_run_multi_async(lsdr_rx *const *rxs, unsigned n_rx, const void *const *ins, size_t n_in, lsdr_softsymbol *const *outs,
                            size_t cap_out, size_t *consumed) {
  LSDR_ARG(rxs && n_rx >= 1 && ins && outs && consumed);
  for (unsigned i = 0; i < n_rx; ++i) {
    LSDR_ARG(rxs[i] && (ins[i] || !n_in) && outs[i]);
    if (rxs[i]->cfg.mode != LSDR_RX_TILED) { lsdr_set_error("cstln_receiver: lsdr_rx_run_multi_async needs LSDR_RX_TILED"); return LSDR_E_UNSUPPORTED; }
    for (unsigned k = 0; k < i; ++k) LSDR_ARG(rxs[k] != rxs[i]);
  }
  return rx_tiled_enqueue_multi(rxs, n_rx, ins, n_in, outs, cap_out, consumed);
}

int lsdr_rx_wait(lsdr_rx *r, size_t *produced) {
  LSDR_ARG(r && produced);
  return rx_tiled_wait(r, produced);
}
float lsdr_rx_retired_freq_tap(const lsdr_rx *r) { return r ? r->retired_freq_tap : 0.f; }

int lsdr_rx_tile_time(lsdr_rx *r, int enable, float *avg_ms, unsigned *launches) {
  LSDR_ARG(r);
  if (enable && !r->tev0[0])
    for (int i = 0; i < lsdr_rx::kRing; ++i) { LSDR_HIP(hipEventCreate(&r->tev0[i])); LSDR_HIP(hipEventCreate(&r->tev1[i])); }
  if (avg_ms) *avg_ms = r->time_n ? (float)(r->time_ms / r->time_n) : 0.f;
  if (launches) *launches = r->time_n;
  r->time_ms = 0; r->time_n = 0;
  r->time_on = enable != 0;
  return LSDR_OK;
}

int lsdr_rx_run(lsdr_rx *r, const void *in, size_t n_in, lsdr_softsymbol *out, size_t cap_out,
                size_t *consumed, size_t *produced, float *freq_out, float *ss_out, float *mer_out,
                size_t meas_cap, size_t *n_meas, lsdr_cf32 *cstln_out, size_t cstln_cap, size_t *n_cstln) {
  LSDR_ARG(r && consumed && produced);
  *consumed = 0; *produced = 0;
  if (n_meas) *n_meas = 0;
  if (n_cstln) *n_cstln = 0;
  const int ra = lsdr_rx_readahead(r);
  if (n_in < (size_t)(kChunk + ra) || cap_out < (size_t)kChunk) return LSDR_OK;
  LSDR_ARG(in && out);
  if (r->cfg.mode == LSDR_RX_TILED)
    return rx_run_tiled(r, in, n_in, out, cap_out, consumed, produced, freq_out, ss_out, mer_out, meas_cap, n_meas, cstln_out, cstln_cap, n_cstln);
  lsdr_ctx *c = r->ctx;
  LSDR_HIP(hipSetDevice(c->device));

  const size_t max_chunks = (n_in - ra) / kChunk;
  // measurement scratch: the reference gates on pipe room (sdr.h:785-788); with
  // NULL pipes there is no gate, so size the scratch for every chunk.
  const bool want_meas = freq_out || ss_out || mer_out;
  size_t need_meas = max_chunks * kChunk / r->cfg.meas_decimation + 2;
  size_t eff_meas_cap = want_meas ? (meas_cap < need_meas ? meas_cap : need_meas) : need_meas;
  size_t need_cstln = max_chunks + 1;
  size_t eff_cstln_cap = cstln_out ? (cstln_cap < need_cstln ? cstln_cap : need_cstln) : need_cstln;
  if (r->meas_cap < need_meas) {
    (void)hipFree(r->d_meas);
    LSDR_HIP(hipMalloc((void **)&r->d_meas, need_meas * sizeof(rx_meas)));
    r->meas_cap = need_meas;
  }
  if (cstln_out && r->cstln_cap < need_cstln) {
    (void)hipFree(r->d_cstln);
    LSDR_HIP(hipMalloc((void **)&r->d_cstln, need_cstln * sizeof(float2)));
    r->cstln_cap = need_cstln;
  }
  int rc = rx_push_state(r);
  if (rc) return rc;

  rx_serial_args a;
  a.in = in;
  a.n_in = n_in;
  a.out = out;
  a.cap_out = cap_out;
  a.state = r->d_state;
  a.meas = r->d_meas; a.meas_cap = eff_meas_cap;
  a.cstln = cstln_out ? r->d_cstln : nullptr; a.cstln_cap = eff_cstln_cap;
  a.counters = r->d_counters;
  rx_fill_consts(r, a.C, a.T);
  a.readahead = ra;
  size_t shmem = (size_t)(kChunk + ra) * sizeof(float2);
#define LSDR_RX_SERIAL(S) do { if (r->cfg.in_format == LSDR_IN_CU8) hipLaunchKernelGGL((k_rx_serial<S, LSDR_IN_CU8>), dim3(1), dim3(64), shmem, c->stream, a); \
                               else hipLaunchKernelGGL((k_rx_serial<S, LSDR_IN_CF32>), dim3(1), dim3(64), shmem, c->stream, a); } while (0)
  switch (r->cfg.sampler) {
    case LSDR_SAMP_NEAREST: LSDR_RX_SERIAL(0); break;
    case LSDR_SAMP_LINEAR: LSDR_RX_SERIAL(1); break;
    default: LSDR_RX_SERIAL(2); break;
  }
#undef LSDR_RX_SERIAL
  LSDR_HIP(hipGetLastError());

  unsigned long long cnt[4];
  LSDR_HIP(hipMemcpyAsync(cnt, r->d_counters, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
  LSDR_HIP(hipMemcpyAsync(&r->st, r->d_state, sizeof(rx_state_dev), hipMemcpyDeviceToHost, c->stream));
  LSDR_HIP(hipStreamSynchronize(c->stream));
  *consumed = cnt[0];
  *produced = cnt[1];
  if (want_meas && cnt[2]) {
    std::vector<rx_meas> m(cnt[2]);
    LSDR_HIP(hipMemcpy(m.data(), r->d_meas, cnt[2] * sizeof(rx_meas), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < cnt[2]; ++i) {
      if (freq_out) freq_out[i] = m[i].freqw / 65536;                   // freq_tap
      if (ss_out) ss_out[i] = sqrtf(m[i].est_insp);                      // sdr.h:910
      if (mer_out) mer_out[i] = m[i].est_ep ? 10 * logf(m[i].est_sp / m[i].est_ep) / logf(10) : 0;  // sdr.h:912
    }
  }
  if (n_meas) *n_meas = want_meas ? cnt[2] : 0;
  if (cstln_out && cnt[3]) LSDR_HIP(hipMemcpy(cstln_out, r->d_cstln, cnt[3] * sizeof(float2), hipMemcpyDeviceToHost));
  if (n_cstln) *n_cstln = cstln_out ? cnt[3] : 0;
  return LSDR_OK;
}

// ---- lane-per-capture exact receiver -----------------------------------------------------------------------------------
struct lsdr_rx_batch {
  lsdr_rx *proto;                 // tables, constants and the initial loop state of one capture
  unsigned n;
  rx_state_dev *d_states;
  const void **d_in; lsdr_softsymbol **d_out; unsigned long long *d_prod;
  std::vector<unsigned long long> h_prod;
};

int lsdr_rx_batch_create(lsdr_ctx *c, const lsdr_rx_cfg *cfg, unsigned n_streams, lsdr_rx_batch **out) {
  LSDR_ARG(c && cfg && out && n_streams >= 1);
  if (cfg->sampler == LSDR_SAMP_FIR) { lsdr_set_error("rx_batch: nearest and linear samplers only"); return LSDR_E_UNSUPPORTED; }
  lsdr_rx_cfg pc = *cfg;
  pc.mode = LSDR_RX_SERIAL;
  lsdr_rx *proto = nullptr;
  int rc = lsdr_rx_create(c, &pc, &proto);
  if (rc) return rc;
  lsdr_rx_batch *b = new lsdr_rx_batch();
  b->proto = proto; b->n = n_streams;
  std::vector<rx_state_dev> init(n_streams, proto->st);
  LSDR_HIP(hipMalloc((void **)&b->d_states, n_streams * sizeof(rx_state_dev)));
  LSDR_HIP(hipMemcpy(b->d_states, init.data(), n_streams * sizeof(rx_state_dev), hipMemcpyHostToDevice));
  LSDR_HIP(hipMalloc((void **)&b->d_in, n_streams * sizeof(void *)));
  LSDR_HIP(hipMalloc((void **)&b->d_out, n_streams * sizeof(void *)));
  LSDR_HIP(hipMalloc((void **)&b->d_prod, n_streams * sizeof(unsigned long long)));
  b->h_prod.resize(n_streams);
  *out = b;
  return LSDR_OK;
}

void lsdr_rx_batch_destroy(lsdr_rx_batch *b) {
  if (!b) return;
  (void)hipStreamSynchronize(b->proto->ctx->stream);
  (void)hipFree(b->d_states); (void)hipFree((void *)b->d_in); (void)hipFree((void *)b->d_out); (void)hipFree(b->d_prod);
  lsdr_rx_destroy(b->proto);
  delete b;
}

int lsdr_rx_batch_run(lsdr_rx_batch *b, const void *const *in_dev, size_t n_in, lsdr_softsymbol *const *out_dev, size_t cap_out,
                      size_t *consumed, size_t *produced) {
  LSDR_ARG(b && in_dev && out_dev && consumed);
  *consumed = 0;
  lsdr_rx *r = b->proto;
  lsdr_ctx *c = r->ctx;
  LSDR_HIP(hipSetDevice(c->device));
  const int ra = lsdr_rx_readahead(r);
  size_t chunks = n_in >= (size_t)(kChunk + ra) ? (n_in - ra) / kChunk : 0;
  const unsigned sym_per_chunk = (unsigned)(kChunk / (r->omega - 0.1f)) + 2;
  if ((size_t)sym_per_chunk * chunks > cap_out) chunks = cap_out / sym_per_chunk;       // every capture runs the same chunks
  if (produced) for (unsigned i = 0; i < b->n; ++i) produced[i] = 0;
  if (!chunks) return LSDR_OK;
  LSDR_HIP(hipMemcpyAsync((void *)b->d_in, in_dev, b->n * sizeof(void *), hipMemcpyHostToDevice, c->stream));
  LSDR_HIP(hipMemcpyAsync((void *)b->d_out, out_dev, b->n * sizeof(void *), hipMemcpyHostToDevice, c->stream));
  LSDR_HIP(hipStreamSynchronize(c->stream));                                                 // (the pointer arrays are the caller's)
  rx_batch_args a;
  a.in = b->d_in; a.out = b->d_out; a.chunks = chunks; a.states = b->d_states; a.produced = b->d_prod; a.n_streams = b->n;
  rx_fill_consts(r, a.C, a.T);
  const unsigned blocks = (b->n + 63) / 64;
#define LSDR_RX_BATCH(S) do { if (r->cfg.in_format == LSDR_IN_CU8) hipLaunchKernelGGL((k_rx_batch<S, LSDR_IN_CU8>), dim3(blocks), dim3(64), 0, c->stream, a); \
                              else hipLaunchKernelGGL((k_rx_batch<S, LSDR_IN_CF32>), dim3(blocks), dim3(64), 0, c->stream, a); } while (0)
  if (r->cfg.sampler == LSDR_SAMP_NEAREST) LSDR_RX_BATCH(0);
  else LSDR_RX_BATCH(1);
#undef LSDR_RX_BATCH
  LSDR_HIP(hipGetLastError());
  LSDR_HIP(hipMemcpyAsync(b->h_prod.data(), b->d_prod, b->n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  LSDR_HIP(hipStreamSynchronize(c->stream));
  if (produced) for (unsigned i = 0; i < b->n; ++i) produced[i] = (size_t)b->h_prod[i];
  *consumed = chunks * kChunk;
  return LSDR_OK;
}

int lsdr_rx_batch_get_state(lsdr_rx_batch *b, unsigned stream, lsdr_rx_state *st) {
  LSDR_ARG(b && st && stream < b->n);
  rx_state_dev s;
  LSDR_HIP(hipMemcpy(&s, b->d_states + stream, sizeof(s), hipMemcpyDeviceToHost));
  rx_state_export(s, st);
  return LSDR_OK;
}

}  // extern "C"

#include "rxb_host.h"
