// leansdr_amd/csrc/hs.hip — the `--hs` path of leandvb (leandvb.cc:727-969) on gfx950:
//
//   fast_qpsk_receiver<u8>  sdr.h:946-1189   all-integer QPSK receiver on cu8 samples (phase-only look-up tables,
//                                            u16 carrier phase, 64-bit frequency word, float symbol clock).
//                                            A decision-feedback recurrence like cstln_receiver: one lane runs it
//                                            in the reference's exact operation order, its three dependent table
//                                            look-ups go through the scalar cache; the other lanes stage samples.
//   dvb_deconvol_sync<u8>   dvb.h:612-707    algebraic deconvolution by 32-bit bit-parallel XORs
//                           convolutional.h:80-192 (deconvol_poly2<…,0x3ba,0x38f70>) with alignment search.
//                                            Every 64-byte chunk is a pure function of its 512 symbols and the 32
//                                            before them; the alignment in force after a resync chunk is the
//                                            arg-min of the four error counts of that chunk (independent of the
//                                            previous one) → two fully parallel kernels.
//
// Tables are built on the host with the reference's libm expressions (host_tables.cpp) and uploaded once.
#include "lsdr_internal.h"

namespace {

constexpr int kChunk = 128;   // fast_qpsk_receiver::chunk_size, sdr.h:957

typedef unsigned hs_u32c __attribute__((address_space(4)));
typedef unsigned short hs_u16c __attribute__((address_space(4)));

struct fq_state {               // members of fast_qpsk_receiver that cross run() calls
  float mu;
  unsigned phase;               // u_angle (low 16 bits)
  long long freqw, min_freqw, max_freqw;
  unsigned long long meas_count;
  unsigned char hist_p[3][2], hist_c[3][2];
};

struct fq_args {
  const unsigned char *in;      // cu8 samples
  unsigned long long n_in;
  unsigned char *out;           // hard symbols
  unsigned long long cap_out;
  fq_state *state;
  float *freq; unsigned long long freq_cap;       // device scratch, may be null
  unsigned char *cstln; unsigned long long cstln_cap;
  unsigned long long *counters;                   // [0] consumed, [1] produced, [2] n_freq, [3] n_cstln
  const unsigned *polar;        // [65536]  a | r << 16, index re*256+im
  const unsigned short *rect;   // [256*256] re | im << 8, index a*256 + r
  const unsigned short *sincos; // [65536]  re | im << 8
  float omega, gain_mu;
  long long freq_alpha, freq_beta;
  unsigned long long meas_decimation;
  int allow_drift;
};

__device__ __forceinline__ unsigned uload32(const unsigned *t, unsigned i) {
  const unsigned si = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
  return ((const hs_u32c *)t)[si];
}
__device__ __forceinline__ unsigned uload16(const unsigned short *t, unsigned i) {
  const unsigned si = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
  return ((const hs_u16c *)t)[si];
}

// One wavefront; lane 0 = the recurrence (sdr.h:997-1140), all lanes = staging.
__global__ __launch_bounds__(64) void k_fastqpsk_serial(fq_args a) {
  __shared__ unsigned short buf[kChunk + 2];      // cu8 samples of the chunk (+1 for interpolation)
  __shared__ fq_state st;
  __shared__ unsigned long long sh_nout, sh_nf, sh_nc;
  const int lane = threadIdx.x;
  if (lane == 0) { st = *a.state; sh_nout = 0; sh_nf = 0; sh_nc = 0; }
  __syncthreads();
  const unsigned short *in16 = reinterpret_cast<const unsigned short *>(a.in);
  const unsigned long long max_meas = kChunk / a.meas_decimation + 1;
  unsigned long long pos = 0;
  while (true) {
    if (a.n_in < pos || a.n_in - pos < (unsigned long long)(kChunk + 1)) break;   // sdr.h:1010-1013
    if (a.cap_out - sh_nout < (unsigned long long)kChunk) break;
    if (a.freq && a.freq_cap - sh_nf < max_meas) break;
    if (a.cstln && a.cstln_cap - sh_nc < max_meas) break;
    for (int k = lane; k < kChunk + 1; k += 64) buf[k] = in16[pos + k];
    __syncthreads();
    if (lane == 0) {
      float mu = st.mu;
      unsigned phase = st.phase & 0xffffu;
      long long freqw = st.freqw;
      unsigned char *po = a.out + sh_nout;
      int cnt = 0;
      unsigned s_re = 0, s_im = 0, symbol_arg = 0;
      for (int n = 0; n < kChunk; ++n) {
        if (mu < 1) {
          const unsigned x0 = buf[n], x1 = buf[n + 1];             // re | im << 8
          const unsigned p0 = uload32(a.polar, (x0 & 255u) * 256u + (x0 >> 8));
          const unsigned p1 = uload32(a.polar, (x1 & 255u) * 256u + (x1 >> 8));
          const unsigned a0 = (((p0 & 0xffffu) - phase) & 0xffffu) >> 8;
          const unsigned a1 = (unsigned)(((long long)(p1 & 0xffffu) - ((long long)phase + freqw)) & 0xffff) >> 8;
          const unsigned r0 = uload16(a.rect, a0 * 256u + ((p0 >> 16) >> 1));
          const unsigned r1 = uload16(a.rect, a1 * 256u + ((p1 >> 16) >> 1));
          const int p0re = (int)(r0 & 255u), p0im = (int)(r0 >> 8), p1re = (int)(r1 & 255u), p1im = (int)(r1 >> 8);
          s_re = (unsigned)(int)((float)p0re + (float)(p1re - p0re) * mu) & 255u;    // (int)(int + int*float) → u8
          s_im = (unsigned)(int)((float)p0im + (float)(p1im - p0im) * mu) & 255u;
          symbol_arg = uload32(a.polar, s_re * 256u + s_im) & 0xffffu;
          // quadrant → symbol {0,2,3,1}, sdr.h:1066-1069
          po[cnt++] = (unsigned char)((0x1320u >> ((symbol_arg >> 14) * 4)) & 15u);
          const long long phase_error = (long long)(int)(symbol_arg & 16383u) - 8192;           // sdr.h:1072
          phase = (unsigned)((long long)phase + ((phase_error * a.freq_alpha + 32768) >> 16)) & 0xffffu;
          freqw += (phase_error * a.freq_beta + 32768 * 256) >> 24;
          st.hist_p[2][0] = st.hist_p[1][0]; st.hist_p[2][1] = st.hist_p[1][1];
          st.hist_c[2][0] = st.hist_c[1][0]; st.hist_c[2][1] = st.hist_c[1][1];
          st.hist_p[1][0] = st.hist_p[0][0]; st.hist_p[1][1] = st.hist_p[0][1];
          st.hist_c[1][0] = st.hist_c[0][0]; st.hist_c[1][1] = st.hist_c[0][1];
          st.hist_p[0][0] = (unsigned char)s_re; st.hist_p[0][1] = (unsigned char)s_im;
          const unsigned c = uload16(a.sincos, ((symbol_arg & 49152u) + 8192u) & 0xffffu);
          st.hist_c[0][0] = (unsigned char)(c & 255u); st.hist_c[0][1] = (unsigned char)(c >> 8);
          const int muerr =
              ((int)(signed char)(st.hist_p[0][0] - st.hist_p[2][0]) * ((int)st.hist_c[1][0] - 128) +
               (int)(signed char)(st.hist_p[0][1] - st.hist_p[2][1]) * ((int)st.hist_c[1][1] - 128)) -
              ((int)(signed char)(st.hist_c[0][0] - st.hist_c[2][0]) * ((int)st.hist_p[1][0] - 128) +
               (int)(signed char)(st.hist_c[0][1] - st.hist_c[2][1]) * ((int)st.hist_p[1][1] - 128));
          float mucorr = (float)muerr * a.gain_mu;
          const float max_mucorr = 0.1f;
          if (mucorr < -max_mucorr) mucorr = -max_mucorr;
          if (mucorr > max_mucorr) mucorr = max_mucorr;
          mu += mucorr;
          mu += a.omega;
        }
        mu = mu - 1;
        phase = (unsigned)((long long)phase + freqw) & 0xffffu;
      }
      sh_nout += (unsigned long long)cnt;
      if (symbol_arg && a.cstln) { a.cstln[2 * sh_nc] = (unsigned char)s_re; a.cstln[2 * sh_nc + 1] = (unsigned char)s_im; ++sh_nc; }
      if (!a.allow_drift)
        if (freqw < st.min_freqw || freqw > st.max_freqw) freqw = (st.max_freqw + st.min_freqw) / 2;
      st.mu = mu; st.phase = phase; st.freqw = freqw;
      st.meas_count += kChunk;
      while (st.meas_count >= a.meas_decimation) {
        st.meas_count -= a.meas_decimation;
        if (a.freq) a.freq[sh_nf++] = (float)freqw / 65536;
      }
    }
    pos += kChunk;
    __syncthreads();
  }
  if (lane == 0) {
    *a.state = st;
    a.counters[0] = pos; a.counters[1] = sh_nout; a.counters[2] = sh_nf; a.counters[3] = sh_nc;
  }
}

// ---------------------------------------------------------------- throughput mode (time-tiled, tolerance)
// Same scheme as LSDR_RX_TILED of cstln_receiver (cstln_receiver.hip, rx_tiling.h): one lane per tile; tile 0 (a block
// of its own) continues exactly from the carried state, tile j ≥ 1 starts `warm_chunks` early with mu = phase = 0,
// history cleared and the carried frequency word; the seam kernels reconcile carrier quadrant (symbols relabelled by
// the accumulated quadrant step), ±1 symbol at tile boundaries and output offsets.  Not bit-exact by construction; the
// transport stream behind it is what the tests compare.
__device__ __forceinline__ unsigned char rx_relabel(unsigned char v, const uint8_t *map) { return map[v]; }
__device__ __forceinline__ unsigned rx_symbol_of(unsigned char v) { return v; }
__device__ __forceinline__ float rx_freq_tap(const fq_state *st) { return (float)st->freqw / 65536.0f / 65536.0f; }   // (not consumed: --hs has no freq_tap user)
__device__ __forceinline__ void rx_rotate_back(fq_state *st, unsigned rot, float quad) {
  st->phase = (st->phase - rot * (unsigned)quad) & 0xffffu;
}
#include "rx_tiling.h"
typedef rx_tile_info_t<unsigned char> fq_tile_info;

struct fq_tiled_args {
  const unsigned char *in;
  unsigned long long total_chunks;
  unsigned first_chunks, tile_chunks, warm_chunks, n_tiles, stage_stride;
  unsigned char *stage;
  unsigned char *wstage; unsigned wstride;   // symbols of each tile's last warm-up chunk (seam vote)
  fq_tile_info *info;
  fq_state *state;
  const unsigned *polar; const unsigned short *rect; const unsigned short *sincos;
  float omega, gain_mu;
  long long freq_alpha, freq_beta;
  unsigned long long meas_decimation;
  int allow_drift;
  long long freq_window;
};

// One 128-sample chunk of fast_qpsk_receiver::run by ONE lane (vector table loads).  Same arithmetic as the serial kernel.
// CLAMP (tolerance tiles only): the frequency word stays within ±a.freq_window of the carried value — a tile that
// starts from scratch next to the PLL's unstable equilibrium can otherwise pump its frequency integrator into a false
// lock that outlasts the tile (seen once in ~3000 tiles at 18 dB: phase ramping through the whole tile).
template <bool CLAMP, typename Emit>
__device__ __forceinline__ int fq_chunk(const fq_tiled_args &a, fq_state &st, const unsigned short *pin, int avail, Emit emit,
                                        long long f_lo = 0, long long f_hi = 0) {
  float mu = st.mu;
  unsigned phase = st.phase & 0xffffu;
  long long freqw = st.freqw;
  int cnt = 0;
  // The tile kernel is bound by the REQUESTS its lanes make (every lane walks its own tile: 64 cache lines per vector-memory instruction, six
  // table look-ups per symbol): the samples therefore come eight at a time — one 16-byte load per ≈ 6 samples instead of two 2-byte loads per
  // symbol — and a symbol whose first sample is the previous symbol's second one (4 of 5 at 1.2 samples per symbol) keeps that look-up.
  typedef unsigned fq_v4u __attribute__((ext_vector_type(4)));
  fq_v4u w = {0u, 0u, 0u, 0u};
  const unsigned sq0 = a.sincos[8192u], sq1 = a.sincos[24576u], sq2 = a.sincos[40960u], sq3 = a.sincos[57344u];
  int wb = -16, pn = -2;
  unsigned pp1 = 0;
  auto sample = [&](int k) -> unsigned {
    const unsigned i = (unsigned)(k - wb);
    const unsigned dw = i < 4u ? (i < 2u ? w.x : w.y) : (i < 6u ? w.z : w.w);
    return (dw >> (16u * (i & 1u))) & 0xffffu;
  };
  for (int n = 0; n < kChunk; ++n) {
    if (mu < 1) {
      if (n + 1 > wb + 7) {
        wb = n & ~1;
        if (wb + 8 <= avail) w = *reinterpret_cast<const fq_v4u *>(pin + wb);
        else {      // the input's last samples: nothing is read behind them (avail = samples readable from pin on)
          unsigned e[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) e[i] = wb + i < avail ? (unsigned)pin[wb + i] : 0u;
          w = (fq_v4u){e[0] | e[1] << 16, e[2] | e[3] << 16, e[4] | e[5] << 16, e[6] | e[7] << 16};
        }
      }
      const unsigned x0 = sample(n), x1 = sample(n + 1);
      const unsigned p0 = n == pn + 1 ? pp1 : a.polar[(x0 & 255u) * 256u + (x0 >> 8)];
      const unsigned p1 = a.polar[(x1 & 255u) * 256u + (x1 >> 8)];
      pn = n; pp1 = p1;
      const unsigned a0 = (((p0 & 0xffffu) - phase) & 0xffffu) >> 8;
      const unsigned a1 = (unsigned)(((long long)(p1 & 0xffffu) - ((long long)phase + freqw)) & 0xffff) >> 8;
      const unsigned r0 = a.rect[a0 * 256u + ((p0 >> 16) >> 1)];
      const unsigned r1 = a.rect[a1 * 256u + ((p1 >> 16) >> 1)];
      const int p0re = (int)(r0 & 255u), p0im = (int)(r0 >> 8), p1re = (int)(r1 & 255u), p1im = (int)(r1 >> 8);
      const unsigned s_re = (unsigned)(int)((float)p0re + (float)(p1re - p0re) * mu) & 255u;
      const unsigned s_im = (unsigned)(int)((float)p0im + (float)(p1im - p0im) * mu) & 255u;
      const unsigned symbol_arg = a.polar[s_re * 256u + s_im] & 0xffffu;
      emit((unsigned char)((0x1320u >> ((symbol_arg >> 14) * 4)) & 15u));
      ++cnt;
      const long long phase_error = (long long)(int)(symbol_arg & 16383u) - 8192;
      phase = (unsigned)((long long)phase + ((phase_error * a.freq_alpha + 32768) >> 16)) & 0xffffu;
      freqw += (phase_error * a.freq_beta + 32768 * 256) >> 24;
      if (CLAMP) { freqw = freqw < f_lo ? f_lo : freqw; freqw = freqw > f_hi ? f_hi : freqw; }
      st.hist_p[2][0] = st.hist_p[1][0]; st.hist_p[2][1] = st.hist_p[1][1];
      st.hist_c[2][0] = st.hist_c[1][0]; st.hist_c[2][1] = st.hist_c[1][1];
      st.hist_p[1][0] = st.hist_p[0][0]; st.hist_p[1][1] = st.hist_p[0][1];
      st.hist_c[1][0] = st.hist_c[0][0]; st.hist_c[1][1] = st.hist_c[0][1];
      st.hist_p[0][0] = (unsigned char)s_re; st.hist_p[0][1] = (unsigned char)s_im;
      const unsigned qd = symbol_arg >> 14;          // sincos[quadrant·16384 + 8192]: four entries in all, fetched once per chunk (sq0 … sq3)
      const unsigned c = qd < 2u ? (qd == 0u ? sq0 : sq1) : (qd == 2u ? sq2 : sq3);
      st.hist_c[0][0] = (unsigned char)(c & 255u); st.hist_c[0][1] = (unsigned char)(c >> 8);
      const int muerr =
          ((int)(signed char)(st.hist_p[0][0] - st.hist_p[2][0]) * ((int)st.hist_c[1][0] - 128) +
           (int)(signed char)(st.hist_p[0][1] - st.hist_p[2][1]) * ((int)st.hist_c[1][1] - 128)) -
          ((int)(signed char)(st.hist_c[0][0] - st.hist_c[2][0]) * ((int)st.hist_p[1][0] - 128) +
           (int)(signed char)(st.hist_c[0][1] - st.hist_c[2][1]) * ((int)st.hist_p[1][1] - 128));
      float mucorr = (float)muerr * a.gain_mu;
      const float max_mucorr = 0.1f;
      if (mucorr < -max_mucorr) mucorr = -max_mucorr;
      if (mucorr > max_mucorr) mucorr = max_mucorr;
      mu += mucorr;
      mu += a.omega;
    }
    mu = mu - 1;
    phase = (unsigned)((long long)phase + freqw) & 0xffffu;
  }
  if (!a.allow_drift)
    if (freqw < st.min_freqw || freqw > st.max_freqw) freqw = (st.max_freqw + st.min_freqw) / 2;
  st.mu = mu; st.phase = phase; st.freqw = freqw;
  return cnt;
}

constexpr int kFqLanes = 64;   // tiles per wavefront (32 until round 6: half the lanes idle; the kernel is bound by its dependent table look-ups per symbol, i.e. by lanes in flight)
__global__ __launch_bounds__(64) void k_fastqpsk_tiles(fq_tiled_args a) {
  unsigned j;
  if (blockIdx.x == 0) { if (threadIdx.x != 0) return; j = 0; }
  else { if (threadIdx.x >= kFqLanes) return; j = 1u + (blockIdx.x - 1u) * kFqLanes + threadIdx.x; }
  if (j >= a.n_tiles) return;
  const unsigned long long first = a.first_chunks, Lc = a.tile_chunks, Wc = a.warm_chunks, total = a.total_chunks;
  unsigned long long cb, c0, c1;
  if (j == 0) { cb = 0; c0 = 0; c1 = first; }
  else { c0 = first + (unsigned long long)(j - 1) * Lc; c1 = c0 + Lc; cb = c0 - Wc; }
  if (c1 > total) c1 = total;
  fq_state s = *a.state;
  fq_tile_info ti;
  ti.has_pre = 0; ti.pre = 0; ti.n_warm = 0; ti.mu_begin = ti.phase_begin = 0.f;
  unsigned char *pw = a.wstage + (unsigned long long)j * a.wstride;
  if (j > 0) {
    s.mu = 0.f; s.phase = 0;
    for (int k = 0; k < 3; ++k) { s.hist_p[k][0] = s.hist_p[k][1] = 0; s.hist_c[k][0] = s.hist_c[k][1] = 0; }
  }
  const long long f_lo = s.freqw - a.freq_window, f_hi = s.freqw + a.freq_window;
  unsigned char last = 0;
  unsigned char *po = a.stage + (unsigned long long)j * a.stage_stride;
  unsigned cnt = 0, got = 0, sacc = 0;
  const unsigned short *in16 = reinterpret_cast<const unsigned short *>(a.in);
  for (unsigned long long c = cb; c < c1; ++c) {
    const bool body = c >= c0;
    if (c == c0) { ti.mu_begin = s.mu; ti.phase_begin = (float)(s.phase & 0xffffu); ti.pre = last; ti.has_pre = got ? 1u : 0u; }
    const bool lastwarm = c + 1 == c0;
    unsigned nw = 0;
    // (body symbols leave four at a time: a quarter of the scattered store requests)
    auto emit = [&](unsigned char v) {
      if (body) {
        sacc |= (unsigned)v << (8u * (cnt & 3u));
        if ((cnt & 3u) == 3u) { *reinterpret_cast<unsigned *>(po + (cnt & ~3u)) = sacc; sacc = 0u; }
        ++cnt;
      } else { last = v; if (lastwarm) pw[nw++] = v; }
    };
    const int avail = (int)((total - c) * kChunk + 1 < 1024 ? (total - c) * kChunk + 1 : 1024);
    const int n = j == 0 ? fq_chunk<false>(a, s, in16 + c * kChunk, avail, emit) : fq_chunk<true>(a, s, in16 + c * kChunk, avail, emit, f_lo, f_hi);
    if (!body) got += (unsigned)n;
    if (lastwarm) ti.n_warm = nw;
  }
  if (cnt & 3u) *reinterpret_cast<unsigned *>(po + (cnt & ~3u)) = sacc;
  ti.mu_end = s.mu; ti.phase_end = (float)(s.phase & 0xffffu); ti.count = cnt;
  a.info[j] = ti;
  if (j == a.n_tiles - 1) {
    s.meas_count = (a.state->meas_count + a.total_chunks * kChunk) % a.meas_decimation;
    *a.state = s;
  }
}

// ---------------------------------------------------------------- dvb_deconvol_sync<u8>
constexpr int kDcBytes = 64, kDcSyms = 512;    // chunk_size bytes ↔ symbols, dvb.h:618
__constant__ unsigned char c_hs_lut[4][4] = {{0, 1, 2, 3}, {2, 0, 3, 1}, {1, 0, 3, 2}, {0, 2, 1, 3}};   // dvb.h:676-699

// One 32-bit output word of deconvol_poly2::run (convolutional.h:101-185): histI/histQ after each of its 32
// symbols are windows of the remapped I/Q bit streams; `prev` = the 32 symbols before the word (may be carried).
__device__ __forceinline__ void hs_word(const unsigned char *sym32, unsigned histI, unsigned histQ, const unsigned char *lut,
                                        unsigned &wd, unsigned &we) {
  const unsigned long long PD = 0x3baull, PE = 0x38f70ull;
  wd = 0; we = 0;
#pragma unroll
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned iq = lut[sym32[31 - bit] & 3u];
    histI = (histI << 1) | (iq >> 1);
    histQ = (histQ << 1) | (iq & 1u);
    if (PD & (2ull << (2 * bit))) wd ^= histI;
    if (PD & (1ull << (2 * bit))) wd ^= histQ;
    if (PE & (2ull << (2 * bit))) we ^= histI;
    if (PE & (1ull << (2 * bit))) we ^= histQ;
  }
}
// history registers of alignment `lut` after the 32 symbols ending just before `p` (p[-32..-1])
__device__ __forceinline__ void hs_hist(const unsigned char *p, const unsigned char *lut, unsigned &hI, unsigned &hQ) {
  hI = 0; hQ = 0;
#pragma unroll
  for (int k = 32; k >= 1; --k) {
    const unsigned iq = lut[p[-k] & 3u];
    hI = (hI << 1) | (iq >> 1);
    hQ = (hQ << 1) | (iq & 1u);
  }
}

struct hsd_args {
  const unsigned char *in;
  unsigned char *out;
  unsigned long long n_chunks;
  int resync_period, resync_phase0, locked0;
  unsigned carryI[4], carryQ[4];   // deconv state of every alignment at chunk 0
  unsigned char *lock_of;          // [n_chunks] alignment in force while chunk c is decoded
  int *errors;                     // [n_resync][4]
};

// (A) error counts of the four alignments on every resync chunk (second half of the chunk's words only,
// convolutional.h:184 — independent of the stale history a non-locked alignment carries).
__global__ __launch_bounds__(256) void k_hsd_errors(hsd_args a, unsigned long long n_resync) {
  const unsigned long long idx = (unsigned long long)blockIdx.x * 256 + threadIdx.x;   // (resync chunk, alignment, word 8..15)
  const unsigned long long r = idx >> 5;
  if (r >= n_resync) return;
  const int s = (int)((idx >> 3) & 3), w = 8 + (int)(idx & 7);
  const unsigned long long first = (unsigned long long)((a.resync_period - a.resync_phase0) % a.resync_period);
  const unsigned long long c = first + r * (unsigned long long)a.resync_period;
  const unsigned char *p = a.in + c * kDcSyms + (unsigned)w * 32u;
  unsigned hI, hQ, wd, we;
  hs_hist(p, c_hs_lut[s], hI, hQ);      // w ≥ 8: the 32 symbols before the word are inside the chunk
  hs_word(p, hI, hQ, c_hs_lut[s], wd, we);
  atomicAdd(a.errors + r * 4 + s, __popc(we));
}

// (B) decode: one thread per output word with the alignment in force for its chunk.
__global__ __launch_bounds__(256) void k_hsd_decode(hsd_args a) {
  const unsigned long long idx = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned long long c = idx >> 4;
  if (c >= a.n_chunks) return;
  const int w = (int)(idx & 15);
  const int s = a.lock_of[c];
  const unsigned char *p = a.in + c * kDcSyms + (unsigned)w * 32u;
  unsigned hI, hQ, wd, we;
  if (c == 0 && w == 0) { hI = a.carryI[s]; hQ = a.carryQ[s]; }
  else hs_hist(p, c_hs_lut[s], hI, hQ);
  hs_word(p, hI, hQ, c_hs_lut[s], wd, we);
  unsigned char *po = a.out + c * kDcBytes + (unsigned)w * 4u;
  po[0] = (unsigned char)(wd >> 24); po[1] = (unsigned char)(wd >> 16); po[2] = (unsigned char)(wd >> 8); po[3] = (unsigned char)wd;
}

}  // namespace

struct lsdr_fastqpsk {
  lsdr_ctx *ctx;
  float omega, min_omega, max_omega, pll_adjustment;
  int allow_drift;
  unsigned long meas_decimation;
  fq_state st;
  bool st_dirty_host;
  fq_state *d_state;
  unsigned *d_polar; unsigned short *d_rect, *d_sincos;
  unsigned long long *d_counters;
  float *d_freq; size_t freq_cap;
  unsigned char *d_cstln; size_t cstln_cap;
  // throughput mode
  int tiled; unsigned tile_len, tile_warmup;
  unsigned char *d_stage; size_t stage_cap;
  unsigned char *d_wstage; size_t wstage_cap;
  fq_tile_info *d_info; rx_tile_fix *d_fix; rx_seam_part *d_part; size_t tiles_cap;
  uint8_t *d_relabel;
  rx_seam_result *h_res, *h_res_dev;
  unsigned last_tiles, last_dup, last_miss, last_bad;
};

struct lsdr_hsdeconv {
  lsdr_ctx *ctx;
  int resync_period, resync_phase, locked;
  unsigned inI[4], inQ[4];
  unsigned char *d_lock; int *d_err; size_t lock_cap, err_cap;
};

static void fq_update_freq_limits(lsdr_fastqpsk *r) {          // sdr.h:987-992
  r->st.min_freqw = (long long)((float)r->st.freqw - 65536 / r->max_omega / 8);
  r->st.max_freqw = (long long)((float)r->st.freqw + 65536 / r->max_omega / 8);
}

extern "C" {

int lsdr_fastqpsk_create(lsdr_ctx *c, float omega, float freq, float pll_adjustment, int allow_drift, unsigned long meas_decimation,
                         lsdr_fastqpsk **out) {
  LSDR_ARG(c && out && omega > 0);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_fastqpsk *r = new lsdr_fastqpsk();
  r->ctx = c;
  r->pll_adjustment = pll_adjustment; r->allow_drift = allow_drift;
  r->meas_decimation = meas_decimation ? meas_decimation : 1048576;
  memset(&r->st, 0, sizeof(r->st));
  const float tol = 10e-6;                                       // set_omega, sdr.h:975-980
  r->omega = omega; r->min_omega = omega * (1 - tol); r->max_omega = omega * (1 + tol);
  r->st.freqw = (long long)(freq * 65536);                       // set_freq, sdr.h:982-985
  fq_update_freq_limits(r);
  if ((long long)(0.0012 * 256 * 65536 / (double)r->omega * (double)r->pll_adjustment) == 0) {
    delete r;
    lsdr_set_error("fast_qpsk_receiver: Excessive oversampling");   // fail() of sdr.h:1002
    return LSDR_E_ARG;
  }
  std::vector<unsigned> polar(65536);
  std::vector<unsigned short> rect(65536), sincos(65536);
  lsdr::build_fastqpsk_tables(polar.data(), rect.data(), sincos.data());
  LSDR_HIP(hipMalloc((void **)&r->d_polar, polar.size() * 4));
  LSDR_HIP(hipMalloc((void **)&r->d_rect, rect.size() * 2));
  LSDR_HIP(hipMalloc((void **)&r->d_sincos, sincos.size() * 2));
  LSDR_HIP(hipMemcpy(r->d_polar, polar.data(), polar.size() * 4, hipMemcpyHostToDevice));
  LSDR_HIP(hipMemcpy(r->d_rect, rect.data(), rect.size() * 2, hipMemcpyHostToDevice));
  LSDR_HIP(hipMemcpy(r->d_sincos, sincos.data(), sincos.size() * 2, hipMemcpyHostToDevice));
  LSDR_HIP(hipMalloc((void **)&r->d_state, sizeof(fq_state)));
  LSDR_HIP(hipMalloc((void **)&r->d_counters, 4 * sizeof(unsigned long long)));
  r->d_freq = nullptr; r->freq_cap = 0; r->d_cstln = nullptr; r->cstln_cap = 0;
  r->tiled = 0; r->tile_len = 0; r->tile_warmup = 0;
  r->d_stage = nullptr; r->stage_cap = 0; r->d_wstage = nullptr; r->wstage_cap = 0; r->d_info = nullptr; r->d_fix = nullptr; r->d_part = nullptr; r->tiles_cap = 0;
  r->d_relabel = nullptr; r->h_res = nullptr; r->h_res_dev = nullptr;
  r->last_tiles = r->last_dup = r->last_miss = r->last_bad = 0;
  r->st_dirty_host = true;
  *out = r;
  return LSDR_OK;
}

void lsdr_fastqpsk_destroy(lsdr_fastqpsk *r) {
  if (!r) return;
  (void)hipStreamSynchronize(r->ctx->stream);
  (void)hipFree(r->d_polar); (void)hipFree(r->d_rect); (void)hipFree(r->d_sincos); (void)hipFree(r->d_state);
  (void)hipFree(r->d_counters); (void)hipFree(r->d_freq); (void)hipFree(r->d_cstln);
  (void)hipFree(r->d_stage); (void)hipFree(r->d_wstage); (void)hipFree(r->d_info); (void)hipFree(r->d_fix); (void)hipFree(r->d_part); (void)hipFree(r->d_relabel);
  if (r->h_res) (void)hipHostFree(r->h_res);
  delete r;
}

int lsdr_fastqpsk_get_state(const lsdr_fastqpsk *r, float *mu, unsigned *phase, long long *freqw, long long *min_freqw,
                            long long *max_freqw) {
  LSDR_ARG(r);
  if (mu) *mu = r->st.mu;
  if (phase) *phase = r->st.phase & 0xffffu;
  if (freqw) *freqw = r->st.freqw;
  if (min_freqw) *min_freqw = r->st.min_freqw;
  if (max_freqw) *max_freqw = r->st.max_freqw;
  return LSDR_OK;
}

int lsdr_fastqpsk_set_tiled(lsdr_fastqpsk *r, int enable, unsigned tile_len, unsigned tile_warmup) {
  LSDR_ARG(r && tile_len % kChunk == 0 && tile_warmup % kChunk == 0);
  r->tiled = enable ? 1 : 0; r->tile_len = tile_len; r->tile_warmup = tile_warmup;
  return LSDR_OK;
}
int lsdr_fastqpsk_tiled_stats(const lsdr_fastqpsk *r, unsigned *tiles, unsigned *dup, unsigned *miss, unsigned *bad_seams) {
  LSDR_ARG(r);
  if (tiles) *tiles = r->last_tiles;
  if (dup) *dup = r->last_dup;
  if (miss) *miss = r->last_miss;
  if (bad_seams) *bad_seams = r->last_bad;
  return LSDR_OK;
}

static int fq_run_tiled(lsdr_fastqpsk *r, const lsdr_cu8 *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *consumed,
                        size_t *produced, const fq_args &sa) {
  lsdr_ctx *c = r->ctx;
  // default warm-up ≈ 400 symbols: the integer loops of this receiver settle more slowly than cstln_receiver's.  TS yield of
  // the 8000-packet chain: 107 symbols → 36 re-syncs, 213 and more → none (tools/chain_bench.py --hs --tiled); but at 213
  // (256 samples at 1.2 samples/symbol) 0.65 % of the seams still lose a symbol that the RS decoder then repairs, at 427
  // (512 samples) none does (bench_more.py c1_hs: 64 698 tiles, no RS correction) — and the longer tiles that go with it
  // are faster (17.4 against 16.0 GS/s).
  unsigned Wc = r->tile_warmup ? r->tile_warmup / kChunk : (unsigned)((400.0f * r->omega + kChunk - 1) / kChunk);
  if (Wc < 1) Wc = 1;
  const unsigned Lc = r->tile_len ? r->tile_len / kChunk : 2 * Wc;
  const unsigned sym_per_chunk = (unsigned)(kChunk / (r->omega - 0.1f)) + 2;   // mu advances by ≥ omega − 0.1 per symbol
  size_t chunks = (n_in - 1) / kChunk;
  if ((size_t)(sym_per_chunk + 1) * chunks > cap_out) chunks = cap_out / (sym_per_chunk + 1);
  if (!chunks) return LSDR_OK;
  const unsigned first = Lc > Wc ? Lc : Wc;
  unsigned n_tiles = 1;
  if (chunks > first) n_tiles += (unsigned)((chunks - first + Lc - 1) / Lc);
  const unsigned stage_stride = ((first > Lc ? first : Lc) * sym_per_chunk + 7u) & ~3u;      // (rows start on four bytes: the tiles store four symbols at a time)
  if (!r->d_relabel) {
    // quadrant step K of a tile's carrier frame against tile 0's: symbol_arg_tile = symbol_arg − K·16384, so the true
    // quadrant is the tile's + K; symbols are quadrant_to_symbol[] = {0,2,3,1} (sdr.h:1067).
    static const unsigned char q2s[4] = {0, 2, 3, 1}, s2q[4] = {0, 3, 1, 2};
    std::vector<uint8_t> rel(4 * 256, 0);
    for (int K = 0; K < 4; ++K) for (int sy = 0; sy < 4; ++sy) rel[K * 256 + sy] = q2s[(s2q[sy] + K) & 3];
    LSDR_HIP(hipMalloc((void **)&r->d_relabel, rel.size()));
    LSDR_HIP(hipMemcpy(r->d_relabel, rel.data(), rel.size(), hipMemcpyHostToDevice));
    LSDR_HIP(hipHostMalloc((void **)&r->h_res, sizeof(rx_seam_result), hipHostMallocDefault));
    LSDR_HIP(hipHostGetDevicePointer((void **)&r->h_res_dev, r->h_res, 0));
  }
  if (r->tiles_cap < n_tiles) {
    (void)hipFree(r->d_info); (void)hipFree(r->d_fix); (void)hipFree(r->d_part);
    LSDR_HIP(hipMalloc((void **)&r->d_info, n_tiles * sizeof(fq_tile_info)));
    LSDR_HIP(hipMalloc((void **)&r->d_fix, n_tiles * sizeof(rx_tile_fix)));
    LSDR_HIP(hipMalloc((void **)&r->d_part, ((n_tiles + kSeamBlock - 1) / kSeamBlock) * sizeof(rx_seam_part)));
    r->tiles_cap = n_tiles;
  }
  if (r->stage_cap < (size_t)n_tiles * stage_stride) {
    (void)hipFree(r->d_stage);
    LSDR_HIP(hipMalloc((void **)&r->d_stage, (size_t)n_tiles * stage_stride));
    r->stage_cap = (size_t)n_tiles * stage_stride;
  }
  if (r->wstage_cap < (size_t)n_tiles * sym_per_chunk) {
    (void)hipFree(r->d_wstage);
    LSDR_HIP(hipMalloc((void **)&r->d_wstage, (size_t)n_tiles * sym_per_chunk));
    r->wstage_cap = (size_t)n_tiles * sym_per_chunk;
  }
  fq_tiled_args a;
  a.in = (const unsigned char *)in; a.total_chunks = chunks;
  a.first_chunks = first; a.tile_chunks = Lc; a.warm_chunks = Wc; a.n_tiles = n_tiles; a.stage_stride = stage_stride;
  a.stage = r->d_stage; a.wstage = r->d_wstage; a.wstride = sym_per_chunk; a.info = r->d_info; a.state = r->d_state;
  a.polar = sa.polar; a.rect = sa.rect; a.sincos = sa.sincos;
  a.omega = sa.omega; a.gain_mu = sa.gain_mu; a.freq_alpha = sa.freq_alpha; a.freq_beta = sa.freq_beta;
  a.meas_decimation = sa.meas_decimation; a.allow_drift = sa.allow_drift;
  a.freq_window = (long long)(65536.0f / r->omega / 2048.0f);
  if (a.freq_window < 8) a.freq_window = 8;
  const unsigned blocks = 1 + (n_tiles - 1 + kFqLanes - 1) / kFqLanes;
  hipLaunchKernelGGL(k_fastqpsk_tiles, dim3(blocks), dim3(64), 0, c->stream, a);
  hipLaunchKernelGGL((k_rx_seam<fq_tile_info, unsigned char>), dim3((n_tiles + kSeamBlock - 1) / kSeamBlock), dim3(kSeamBlock), 0, c->stream,
                     (const fq_tile_info *)r->d_info, r->d_fix, n_tiles, r->omega, 4, 16384.0f, r->d_part,
                     (const unsigned char *)r->d_stage, stage_stride, (const unsigned char *)r->d_wstage, sym_per_chunk,
                     (const uint8_t *)r->d_relabel);
  hipLaunchKernelGGL((k_rx_compact<unsigned char, fq_state>), dim3((n_tiles + kCompactTiles - 1) / kCompactTiles), dim3(64), 0, c->stream,
                     (const unsigned char *)r->d_stage, stage_stride, (const fq_tile_info *)r->d_info,
                     (const rx_tile_fix *)r->d_fix, (const rx_seam_part *)r->d_part, (const uint8_t *)r->d_relabel, n_tiles, 4,
                     16384.0f, out, r->d_state, r->h_res_dev);
  LSDR_HIP(hipGetLastError());
  LSDR_HIP(hipMemcpyAsync(&r->st, r->d_state, sizeof(fq_state), hipMemcpyDeviceToHost, c->stream));
  LSDR_HIP(hipStreamSynchronize(c->stream));
  static const char *const fq_debug = getenv("LSDR_FQ_DEBUG");
  if (const char *e = fq_debug) {
    const unsigned t0 = (unsigned)atoi(e);
    std::vector<fq_tile_info> hi(n_tiles); std::vector<rx_tile_fix> hf(n_tiles);
    LSDR_HIP(hipMemcpy(hi.data(), r->d_info, n_tiles * sizeof(fq_tile_info), hipMemcpyDeviceToHost));
    LSDR_HIP(hipMemcpy(hf.data(), r->d_fix, n_tiles * sizeof(rx_tile_fix), hipMemcpyDeviceToHost));
    for (unsigned t = t0; t < t0 + 6 && t < n_tiles; ++t)
      fprintf(stderr, "tile %u: mu %.3f..%.3f phase %.0f..%.0f count %u n_warm %u has_pre %u | off %llu rot %u drop %u ins %u\n", t,
              hi[t].mu_begin, hi[t].mu_end, hi[t].phase_begin, hi[t].phase_end, hi[t].count, hi[t].n_warm, hi[t].has_pre,
              hf[t].out_offset, hf[t].rot, hf[t].drop_first, hf[t].insert_pre);
  }
  r->last_tiles = n_tiles; r->last_dup = r->h_res->ndup; r->last_miss = r->h_res->nmiss; r->last_bad = r->h_res->nbad;
  *consumed = chunks * kChunk;
  *produced = (size_t)r->h_res->total;
  return LSDR_OK;
}

int lsdr_fastqpsk_run(lsdr_fastqpsk *r, const lsdr_cu8 *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *consumed,
                      size_t *produced, float *freq_out_host, size_t freq_cap, size_t *n_freq, lsdr_cu8 *cstln_out_host,
                      size_t cstln_cap, size_t *n_cstln) {
  LSDR_ARG(r && consumed && produced);
  *consumed = 0; *produced = 0;
  if (n_freq) *n_freq = 0;
  if (n_cstln) *n_cstln = 0;
  if (n_in < (size_t)(kChunk + 1) || cap_out < (size_t)kChunk) return LSDR_OK;
  LSDR_ARG(in && out);
  lsdr_ctx *c = r->ctx;
  LSDR_HIP(hipSetDevice(c->device));
  const size_t max_chunks = n_in / kChunk;
  const size_t need = max_chunks * kChunk / r->meas_decimation + 2, need_c = max_chunks + 1;
  if (freq_out_host && r->freq_cap < need) {
    (void)hipFree(r->d_freq);
    LSDR_HIP(hipMalloc((void **)&r->d_freq, need * sizeof(float)));
    r->freq_cap = need;
  }
  if (cstln_out_host && r->cstln_cap < need_c) {
    (void)hipFree(r->d_cstln);
    LSDR_HIP(hipMalloc((void **)&r->d_cstln, need_c * 2));
    r->cstln_cap = need_c;
  }
  if (r->st_dirty_host) {
    LSDR_HIP(hipMemcpyAsync(r->d_state, &r->st, sizeof(fq_state), hipMemcpyHostToDevice, c->stream));
    LSDR_HIP(hipStreamSynchronize(c->stream));
    r->st_dirty_host = false;
  }
  fq_args a;
  a.in = (const unsigned char *)in; a.n_in = n_in; a.out = out; a.cap_out = cap_out;
  a.state = r->d_state;
  a.freq = freq_out_host ? r->d_freq : nullptr; a.freq_cap = freq_out_host ? (freq_cap < need ? freq_cap : need) : 0;
  a.cstln = cstln_out_host ? r->d_cstln : nullptr; a.cstln_cap = cstln_out_host ? (cstln_cap < need_c ? cstln_cap : need_c) : 0;
  a.counters = r->d_counters;
  a.polar = r->d_polar; a.rect = r->d_rect; a.sincos = r->d_sincos;
  a.omega = r->omega;
  a.gain_mu = (float)(0.02 / (double)(75.0f * 75.0f) * 2);                                            // sdr.h:1004
  a.freq_alpha = (long long)(0.04 * 65536);                                                            // sdr.h:999
  a.freq_beta = (long long)(0.0012 * 256 * 65536 / (double)r->omega * (double)r->pll_adjustment);      // sdr.h:1000
  a.meas_decimation = r->meas_decimation;
  a.allow_drift = r->allow_drift;
  if (r->tiled) return fq_run_tiled(r, in, n_in, out, cap_out, consumed, produced, a);   // (no FREQ / constellation reports)
  hipLaunchKernelGGL(k_fastqpsk_serial, dim3(1), dim3(64), 0, c->stream, a);
  LSDR_HIP(hipGetLastError());
  unsigned long long cnt[4];
  LSDR_HIP(hipMemcpyAsync(cnt, r->d_counters, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
  LSDR_HIP(hipMemcpyAsync(&r->st, r->d_state, sizeof(fq_state), hipMemcpyDeviceToHost, c->stream));
  LSDR_HIP(hipStreamSynchronize(c->stream));
  if (freq_out_host && cnt[2]) LSDR_HIP(hipMemcpy(freq_out_host, r->d_freq, cnt[2] * sizeof(float), hipMemcpyDeviceToHost));
  if (cstln_out_host && cnt[3]) LSDR_HIP(hipMemcpy(cstln_out_host, r->d_cstln, cnt[3] * 2, hipMemcpyDeviceToHost));
  *consumed = (size_t)cnt[0]; *produced = (size_t)cnt[1];
  if (n_freq) *n_freq = (size_t)cnt[2];
  if (n_cstln) *n_cstln = (size_t)cnt[3];
  return LSDR_OK;
}

// ---------------------------------------------------------------- dvb_deconvol_sync<u8>
int lsdr_hsdeconv_create(lsdr_ctx *c, int resync_period, lsdr_hsdeconv **out) {
  LSDR_ARG(c && out && resync_period >= 1);
  lsdr_hsdeconv *d = new lsdr_hsdeconv();
  d->ctx = c; d->resync_period = resync_period; d->resync_phase = 0; d->locked = 0;
  memset(d->inI, 0, sizeof(d->inI)); memset(d->inQ, 0, sizeof(d->inQ));
  d->d_lock = nullptr; d->d_err = nullptr; d->lock_cap = 0; d->err_cap = 0;
  *out = d;
  return LSDR_OK;
}
void lsdr_hsdeconv_destroy(lsdr_hsdeconv *d) {
  if (!d) return;
  (void)hipStreamSynchronize(d->ctx->stream);
  (void)hipFree(d->d_lock); (void)hipFree(d->d_err);
  delete d;
}
int lsdr_hsdeconv_locked(const lsdr_hsdeconv *d) { return d ? d->locked : 0; }

int lsdr_hsdeconv_run(lsdr_hsdeconv *d, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out, size_t *consumed,
                      size_t *produced) {
  LSDR_ARG(d && consumed && produced);
  *consumed = 0; *produced = 0;
  size_t chunks = n_in / kDcSyms;                      // while in.readable() >= 512 && out.writable() >= 64, dvb.h:636-637
  if (chunks > cap_out / kDcBytes) chunks = cap_out / kDcBytes;
  if (!chunks) return LSDR_OK;
  LSDR_ARG(in && out);
  lsdr_ctx *c = d->ctx;
  LSDR_HIP(hipSetDevice(c->device));
  const int P = d->resync_period, ph0 = d->resync_phase;
  const size_t first = (size_t)((P - ph0) % P);
  const size_t n_resync = first < chunks ? (chunks - first + P - 1) / P : 0;
  if (d->lock_cap < chunks) {
    (void)hipFree(d->d_lock);
    LSDR_HIP(hipMalloc((void **)&d->d_lock, chunks));
    d->lock_cap = chunks;
  }
  if (d->err_cap < n_resync * 4 + 4) {
    (void)hipFree(d->d_err);
    LSDR_HIP(hipMalloc((void **)&d->d_err, (n_resync * 4 + 4) * sizeof(int)));
    d->err_cap = n_resync * 4 + 4;
  }
  hsd_args a;
  a.in = in; a.out = out; a.n_chunks = chunks;
  a.resync_period = P; a.resync_phase0 = ph0; a.locked0 = d->locked;
  for (int s = 0; s < 4; ++s) { a.carryI[s] = d->inI[s]; a.carryQ[s] = d->inQ[s]; }
  a.lock_of = d->d_lock; a.errors = d->d_err;
  std::vector<int> err(n_resync * 4);
  if (n_resync) {
    LSDR_HIP(hipMemsetAsync(d->d_err, 0, n_resync * 4 * sizeof(int), c->stream));
    hipLaunchKernelGGL(k_hsd_errors, dim3((unsigned)((n_resync * 32 + 255) / 256)), dim3(256), 0, c->stream, a,
                       (unsigned long long)n_resync);
    LSDR_HIP(hipGetLastError());
    LSDR_TRY(lsdr_stage_d2h(c, err.data(), d->d_err, err.size() * sizeof(int)));
    LSDR_TRY(lsdr_stage_sync(c));
  }
  // alignment in force per chunk: `locked` changes after a resync chunk to the first arg-min of its error counts
  std::vector<unsigned char> lock_of(chunks);
  int locked = d->locked;
  size_t r = 0;
  for (size_t cc = 0; cc < chunks; ++cc) {
    lock_of[cc] = (unsigned char)locked;
    if (r < n_resync && cc == first + r * (size_t)P) {
      int best = 0, eb = 1 << 30;
      for (int s = 0; s < 4; ++s) if (err[r * 4 + s] < eb) { eb = err[r * 4 + s]; best = s; }
      locked = best;
      ++r;
    }
  }
  LSDR_TRY(lsdr_stage_h2d(c, d->d_lock, lock_of.data(), chunks));
  hipLaunchKernelGGL(k_hsd_decode, dim3((unsigned)((chunks * 16 + 255) / 256)), dim3(256), 0, c->stream, a);
  LSDR_HIP(hipGetLastError());
  // carried deconv state (inI/inQ) of every alignment = remapped last 32 symbols of the last chunk IT processed:
  // the locked one processed every chunk, the others the last resync chunk of this call (if any).
  std::vector<unsigned char> tail(32);
  auto hist_of = [&](size_t chunk, int s) -> int {
    LSDR_HIP(hipMemcpyAsync(tail.data(), in + (chunk + 1) * kDcSyms - 32, 32, hipMemcpyDeviceToHost, c->stream));
    LSDR_HIP(hipStreamSynchronize(c->stream));
    static const unsigned char luts[4][4] = {{0, 1, 2, 3}, {2, 0, 3, 1}, {1, 0, 3, 2}, {0, 2, 1, 3}};
    unsigned hI = 0, hQ = 0;
    for (int k = 0; k < 32; ++k) { const unsigned iq = luts[s][tail[k] & 3]; hI = (hI << 1) | (iq >> 1); hQ = (hQ << 1) | (iq & 1); }
    d->inI[s] = hI; d->inQ[s] = hQ;
    return LSDR_OK;
  };
  // Which chunk did each alignment process last?  Alignment s processes chunk cc iff cc is a resync chunk or s == lock_of[cc].
  for (int s = 0; s < 4; ++s) {
    long long last = -1;
    for (long long cc = (long long)chunks - 1; cc >= 0; --cc) {
      const bool resync = (size_t)cc >= first && ((size_t)cc - first) % (size_t)P == 0;
      if (resync || lock_of[cc] == s) { last = cc; break; }
    }
    if (last >= 0) { int rc = hist_of((size_t)last, s); if (rc) return rc; }
  }
  LSDR_TRY(lsdr_stage_sync(c));
  d->locked = locked;
  d->resync_phase = (int)(((size_t)ph0 + chunks) % (size_t)P);
  *consumed = chunks * kDcSyms;
  *produced = chunks * kDcBytes;
  return LSDR_OK;
}

}  // extern "C"
