// leansdr_amd/csrc/rxb_device.h — device side of the CAPTURE BATCH receiver (lsdr_capture_batch, include/lsdr_hip.h): the front end of
// leandvb's default graph for `--u8` input (leandvb.cc:211-217,296-301,476-510: cconverter<u8> → auto_notch(1 slot) → cstln_receiver with
// the linear sampler), for B independent captures decoded from their first sample, all of them in ONE set of launches.  Included inside
// cstln_receiver.hip's anonymous namespace (it reuses rx_tile_exact, the seam pass and the packed compaction of rx_tiling.h).
//
//   k_rxb_detect_fft / k_rxb_detect_peaks / k_rxb_iv   auto_notch::detect() (sdr.h:76-118) at every detect point of every capture: the
//                        reference's FFT bit for bit (notch_detect.h), first maximum, and per detect interval the notch's pole
//                        p = (1−k)·exp(j2π·bin/4096) and whether the estimator restarts there (sdr.h:94-103: only when the bin changes)
//   k_rxb_notch_pre      zero-start sums of the estimator recurrence over blocks of `pre_block` samples: a tile that starts in the
//                        middle of a capture gets its estimator from the few blocks in front of it ((1−k)^8192 < 1e-7)
//   k_rxb_tiles<NOTCH>   the tolerance tiles (one lane per tile, 64 consecutive tiles of one capture per wavefront, cu8 samples staged
//                        through LDS by buffer→LDS loads), packed 2-bit decisions out (rx_tiling.h "hs2")
//   k_rxb_seam / k_rxb_compact   rx_tiling.h's seam pass and packed compaction, blockIdx.y = capture
//
// The notch inside a tile.  sdr.h:119-138 with one slot is, per detect interval, estim[n] = (1−k)·estim[n−1] + k·x[n]·conj(e[n]),
// out[n] = x[n] − estim[n]·e[n], e[n] = exp(j2π·bin·n/4096) (n counted from the block start; 4096·bin/4096 is whole, so the phasor runs
// on across blocks).  With S[n] = estim[n]·e[n]:  S[n] = p·S[n−1] + k·x[n],  out[n] = x[n] − S[n] — one complex multiply-add per sample,
// no table.  TOLERANCE MODE like lsdr_notch_fir (float32 with exact phases, not the reference's table of cosf/sinf of a rounded angle):
// tests/test_gpu_capture_batch.py holds the notched samples against the oracle's auto_notch, and the TS against the reference binary's.
//
// The tile's symbol step.  Same loop as rx_tile_tol (sdr.h:790-916 per chunk: interpolate, AGC, slice, PLL, Mueller & Müller, estimators
// per chunk) restated for QPSK / linear sampler / packed decisions with about half the vector instructions per symbol — this kernel is
// bound by VALU issue, not by HBM:
//   * the linear sampler's two derotations (sdr.h:614-623) as ONE: s = (p0·(1−mu) + p1·e^{−jf}·mu)·e^{−j·phase}, e^{−jf} once per chunk
//     (sampler->update_freq is per chunk, sdr.h:790), e^{−j·phase} from v_cos/v_sin on the unquantised phase;
//   * decisions by arithmetic on the truncated coordinates (what cstln_lut<256> holds for QPSK, sdr.h:529-560): symbol = the two sign
//     bits, phase error = atan((|Q|−|I|)/(|Q|+|I|)) by an odd polynomial — atan2 − π/4 without the octant folding;
//   * fused multiply-adds throughout; the sample-skipping steps as one floor();
//   * the samples of the next symbol are fetched from LDS a whole symbol step ahead.
#ifndef LSDR_RXB_DEVICE_H
#define LSDR_RXB_DEVICE_H

struct rxb_iv { float pr, pi, k; int seg_block, bin, pad0, pad1, pad2; };   // one detect interval: pole, gain (0: no notch yet), first pre-block of its constant-bin segment, bin

struct rxb_cap {
  const unsigned char *in;             // cu8 items
  unsigned long long total_chunks;     // 128-sample chunks the receiver runs over
  unsigned n_tiles, n_det;
  unsigned *hstage; unsigned long long hpitch;
  rx_tile_info_h *hinfo;
  rx_tile_fix *fix; rx_seam_part *part;
  unsigned *out_words;
  rx_seam_result *res;                 // device memory: total symbols, seam statistics
  rx_state_dev *state_end;             // where the last tile leaves phase / freqw (rx_tiling.h reads freq_tap there)
  rx_ema_map *ema_scratch;             // [2]: what rx_tile_exact writes for the estimator scan nobody runs here
  rxb_iv *iv;                          // [n_det + 1]
  float2 *T;                           // [n_pre] zero-start block sums of S
  int *cand;                           // [n_det][kDetMaxSlots]
  float2 *halves;                      // [n_det][2][2048]
};

struct rxb_args {
  const rxb_cap *caps;
  const unsigned *iv_of_block;         // [blocks of 4096 samples] detect interval a block belongs to (the same for every capture)
  const unsigned *det_block;           // [n_det] block index of every detect point
  const float2 *om;                    // reverse-FFT twiddles (notch_detect.h)
  const rx_state_dev *state0;          // loop state right after construction (every capture starts there)
  unsigned tile_chunks, warm_chunks;
  unsigned pre_block, pre_look;        // samples per pre-pass block; blocks a tile looks back
  float nk, l2omk;                     // auto_notch::k, log2(1 − k)
  rx_consts C;
  rx_tables T;
};

// ---- detect chain ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rxb_detect_fft(rxb_args A) {
  const rxb_cap &cap = A.caps[blockIdx.y];
  if ((blockIdx.x >> 1) >= cap.n_det) return;
  const unsigned char *src = cap.in + 2ull * kDetN * A.det_block[blockIdx.x >> 1];
  cfft_half_body_t([&](unsigned i) { const uchar2 v = reinterpret_cast<const uchar2 *>(src)[i]; return cu8_to_cf32(v.x, v.y); }, A.om, cap.halves,
                   blockIdx.x);
}
__global__ __launch_bounds__(256) void k_rxb_detect_peaks(rxb_args A) {
  const rxb_cap &cap = A.caps[blockIdx.y];
  if (blockIdx.x >= cap.n_det) return;
  notch_peaks_body(cap.halves, A.om, (float)(1.0 / kDetN), 1, cap.cand, blockIdx.x);
}
// (1−k)^m·exp(j2π·bin·m/4096): exact angle reduction in integers, the magnitude through exp2
__device__ __forceinline__ float2 rxb_ppow(int bin, float l2omk, unsigned m) {
  const float mag = __builtin_exp2f(l2omk * (float)m);
  const float rev = (float)(((unsigned)bin * m) & 4095u) * (1.0f / 4096.0f);
  return make_float2(mag * __builtin_amdgcn_cosf(rev), mag * __builtin_amdgcn_sinf(rev));
}
// one thread per capture: the detect intervals' notch parameters (sdr.h:94-103: a slot restarts only when its bin changes)
__global__ __launch_bounds__(64) void k_rxb_iv(rxb_args A, unsigned n_caps) {
  const unsigned c = blockIdx.x * 64u + threadIdx.x;
  if (c >= n_caps) return;
  const rxb_cap &cap = A.caps[c];
  rxb_iv v; v.pr = v.pi = v.k = 0.f; v.seg_block = 0; v.bin = -1; v.pad0 = v.pad1 = v.pad2 = 0;
  cap.iv[0] = v;
  int bin_prev = -1;
  for (unsigned q = 0; q < cap.n_det; ++q) {
    const int bin = cap.cand[q * kDetMaxSlots];
    if (bin != bin_prev) v.seg_block = (int)(A.det_block[q] * (kDetN / A.pre_block));
    const double a = 2.0 * M_PI * (double)bin / kDetN, omk = 1.0 - (double)A.nk;
    v.pr = (float)(omk * cos(a)); v.pi = (float)(omk * sin(a)); v.k = A.nk; v.bin = bin;
    cap.iv[q + 1] = v;
    bin_prev = bin;
  }
}

// ---- estimator pre-pass ------------------------------------------------------------------------------------------------------------
// T[b] = Σ_{i in block b} k·p^(end−1−i)·(x[i]−128): what S is right behind block b if it was 0 in front of it.  One workgroup of 256 per
// block, 16 consecutive samples per thread (Horner), the threads' partial sums weighted by p^(16·(255−t)) and added up.
__global__ __launch_bounds__(256) void k_rxb_notch_pre(rxb_args A) {
  const rxb_cap &cap = A.caps[blockIdx.y];
  const unsigned PB = A.pre_block, per = PB / 256u;
  const unsigned long long pos = (unsigned long long)blockIdx.x * PB;
  if (pos + PB > cap.total_chunks * kChunk + 1) return;           // (only blocks in front of a tile start are ever read)
  const rxb_iv v = cap.iv[A.iv_of_block[pos >> 12]];
  __shared__ float2 red[4];
  float2 acc = make_float2(0.f, 0.f);
  if (v.k != 0.f) {
    const unsigned char *src = cap.in + 2 * (pos + (unsigned long long)threadIdx.x * per);
    for (unsigned i = 0; i < per; i += 8) {
      const uint4 w = *reinterpret_cast<const uint4 *>(src + 2 * i);   // 8 samples (a capture buffer is 16-byte aligned, blocks are multiples of 1024 samples)
      const unsigned ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned s = ws[q] >> (16 * h);
          const float xr = (float)((int)(s & 255u) - 128), xi = (float)((int)((s >> 8) & 255u) - 128);
          const float nr = __builtin_fmaf(v.pr, acc.x, __builtin_fmaf(-v.pi, acc.y, v.k * xr));
          const float ni = __builtin_fmaf(v.pr, acc.y, __builtin_fmaf(v.pi, acc.x, v.k * xi));
          acc.x = nr; acc.y = ni;
        }
      }
    }
    const float2 wgt = rxb_ppow(v.bin, A.l2omk, per * (255u - threadIdx.x));     // this thread's span ends that many samples in front of the block end
    acc = make_float2(acc.x * wgt.x - acc.y * wgt.y, acc.x * wgt.y + acc.y * wgt.x);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { acc.x += __shfl_xor(acc.x, d, 64); acc.y += __shfl_xor(acc.y, d, 64); }
  if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    cap.T[blockIdx.x] = make_float2(red[0].x + red[1].x + red[2].x + red[3].x, red[0].y + red[1].y + red[2].y + red[3].y);
}

// ---- tiles -------------------------------------------------------------------------------------------------------------------------
// atan(u)·65536/2π for |u| ≤ 1, odd polynomial (max error 2e-6 rad = 0.02 table units)
__host__ __device__ __forceinline__ float rxb_atan_units(float u) {
  const float S = 10430.3784f;
  const float u2 = u * u;
  float p = -0.0117212f * S;
  p = __builtin_fmaf(p, u2, 0.05265332f * S);
  p = __builtin_fmaf(p, u2, -0.11643287f * S);
  p = __builtin_fmaf(p, u2, 0.19354346f * S);
  p = __builtin_fmaf(p, u2, -0.33262347f * S);
  p = __builtin_fmaf(p, u2, 0.99997726f * S);
  return p * u;
}
// the table's phase_error for the truncated coordinates (Ii, Qi), as a float (an integer value): sdr.h:548-556 for QPSK
__host__ __device__ __forceinline__ float rxb_phase_error(int Ii, int Qi) {
  const float a = __builtin_fabsf((float)Ii) + 1e-6f, b = __builtin_fabsf((float)Qi);   // (+1e-6: (0,0) gives u = −1, the table's atan2f(0,0) = 0)
#ifdef __HIP_DEVICE_COMPILE__
  const float u = (b - a) * __builtin_amdgcn_rcpf(b + a);
#else
  const float u = (b - a) / (b + a);
#endif
  const float at = rxb_atan_units(u);
  unsigned bits;
  __builtin_memcpy(&bits, &at, 4);
  bits ^= (unsigned)(Ii ^ Qi) & 0x80000000u;              // −pe where exactly one coordinate is negative
  float pe;
  __builtin_memcpy(&pe, &bits, 4);
  return __builtin_truncf(pe);                            // the table's (s32) cast
}

__device__ __forceinline__ float rxb_u8f(unsigned w, int byte) {       // (float)((int)u8 − 128): flip the top bit, sign-extend
  return (float)(int)(signed char)((w ^ 0x8080u) >> (8 * byte));
}

// ---- the notch's state for a lane that starts at sample `ws` (a multiple of pre_block) of its capture ----------------------------------
struct rxb_notch { float pr, pi, k, sr, si; unsigned ivm; };
__device__ __forceinline__ rxb_notch rxb_notch_start(const rxb_args &A, const rxb_cap &cap, unsigned long long ws) {
  rxb_notch N; N.pr = N.pi = N.k = N.sr = N.si = 0.f;
  N.ivm = A.iv_of_block[ws >> 12];
  const rxb_iv v = cap.iv[N.ivm];
  N.pr = v.pr; N.pi = v.pi; N.k = v.k;
  if (N.k != 0.f) {
    const long long b = (long long)(ws / A.pre_block);
    const float2 ppb = rxb_ppow(v.bin, A.l2omk, A.pre_block);             // p^pre_block
    for (int q = (int)A.pre_look - 1; q >= 0; --q) {
      const long long idx = b - 1 - q;
      if (idx < (long long)v.seg_block || idx < 0) continue;
      const float2 t = cap.T[idx];
      const float nr = N.sr * ppb.x - N.si * ppb.y + t.x, ni = N.sr * ppb.y + N.si * ppb.x + t.y;
      N.sr = nr; N.si = ni;
    }
  }
  return N;
}
// … at the first sample `pos` of a 4096-sample block: a detect point may start another interval there (sdr.h:64-75, 94-103)
__device__ __forceinline__ void rxb_notch_block(const rxb_args &A, const rxb_cap &cap, unsigned long long pos, rxb_notch &N) {
  const unsigned m = A.iv_of_block[pos >> 12];
  if (m == N.ivm) return;
  const rxb_iv v = cap.iv[m];
  if ((unsigned long long)v.seg_block * A.pre_block == pos) { N.sr = 0.f; N.si = 0.f; }     // the bin changed: the estimator restarts (sdr.h:99-101)
  N.pr = v.pr; N.pi = v.pi; N.k = v.k; N.ivm = m;
}
// S[n] = p·S[n−1] + k·x[n]
__device__ __forceinline__ void rxb_notch_step(float pr, float pi, float k, float sr, float si, float xr, float xi, float &tr, float &ti) {
  tr = __builtin_fmaf(pr, sr, __builtin_fmaf(-pi, si, k * xr));
  ti = __builtin_fmaf(pr, si, __builtin_fmaf(pi, sr, k * xi));
}
// Test kernel (lsdr_capture_batch_notched): the notched stream the tiles see, written out — one LANE per pre_block samples, the same
// start state, interval switches and recurrence as rxb_tile.
__global__ __launch_bounds__(64) void k_rxb_notch_dump(rxb_args A, unsigned cap_index, unsigned long long n_samples, float2 *out) {
  const rxb_cap &cap = A.caps[cap_index];
  const unsigned long long seg = (unsigned long long)blockIdx.x * 64u + threadIdx.x, ws = seg * A.pre_block;
  if (ws >= n_samples) return;
  rxb_notch N = rxb_notch_start(A, cap, ws);
  for (unsigned long long i = ws; i < ws + A.pre_block && i < n_samples; ++i) {
    if ((i & 4095ull) == 0) rxb_notch_block(A, cap, i, N);
    const unsigned w = reinterpret_cast<const unsigned short *>(cap.in)[i];
    const float xr = rxb_u8f(w, 0), xi = rxb_u8f(w, 1);
    float tr, ti;
    rxb_notch_step(N.pr, N.pi, N.k, N.sr, N.si, xr, xi, tr, ti);
    N.sr = tr; N.si = ti;
    out[i] = make_float2(xr - tr, xi - ti);
  }
}

// LDS staging of the lean tiles: 32 samples per stage + 16 of look-ahead (the next symbol's pair is read up to 3 samples ahead, a timing
// excursion walks up to omega + 2 ≤ 10): 112-byte rows, 7 KiB per wavefront — 22 wavefronts per CU where the 64-sample stages of
// rx_tile_tol (11 KiB) allow 14; this kernel lives on wavefronts per SIMD (VALU issue, a dependent chain per symbol).
struct rxb_stage {
  static constexpr int kStage = 32, kMargin = 16, kRowBytes = 2 * (kStage + kMargin) + 16;
};
static_assert(rxb_stage::kRowBytes % 16 == 0 && kChunk % rxb_stage::kStage == 0, "stage geometry");

template <bool NOTCH>
__device__ __forceinline__ void rxb_tile(const rxb_args &A, const rxb_cap &cap, unsigned j0, int lane, char *lds) {
  typedef rxb_stage ST;
  constexpr int kStage = ST::kStage, kRowBytes = ST::kRowBytes, kStageLoads = ST::kRowBytes / 16;
  const bool valid = j0 + (unsigned)lane < cap.n_tiles;
  const unsigned j = valid ? j0 + (unsigned)lane : j0;
  const unsigned Lc = A.tile_chunks, Wc = A.warm_chunks;
  const unsigned long long cb = (unsigned long long)(j - 1) * Lc;          // first warm-up chunk; the body starts at chunk cb + Wc
  unsigned long long c1 = cb + Wc + Lc;
  if (c1 > cap.total_chunks) c1 = cap.total_chunks;
  const int nwarm = (int)Wc, nchunks = valid ? (int)(c1 - cb) : 0;
  const int wave_chunks = (int)(Wc + Lc);

  // LDS staging (rx_tile_tol's scheme): one row per lane, kStageLoads buffer→LDS loads of 1 KiB per 64-sample stage
  const unsigned long long addr = (unsigned long long)cap.in;
  const int delta = (int)(addr & 15ull);
  const unsigned long long cb0 = (unsigned long long)(j0 - 1) * Lc;         // first tile of the wavefront
  const unsigned long long bytes = ((cap.total_chunks * kChunk + 1ull) * 2ull + (unsigned)delta + 15ull) & ~15ull;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(cap.in) - delta, 0,
                                                                         (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
  unsigned src_off[kStageLoads];
#pragma unroll
  for (int q = 0; q < kStageLoads; ++q) {
    const unsigned f = (unsigned)q * 1024u + (unsigned)lane * 16u, r = f / (unsigned)kRowBytes, col = f - r * (unsigned)kRowBytes;
    const unsigned long long tile_byte = (cb0 + (unsigned long long)r * Lc) * (kChunk * 2ull);
    src_off[q] = (j0 + r < cap.n_tiles && tile_byte + col < 0xfff00000ull) ? (unsigned)(tile_byte + col) : 0xfffffff0u;
  }
  const char *const row = lds + lane * kRowBytes + delta;                   // sample s0 of the stage in flight sits here

  const rx_consts &C = A.C;
  const rx_state_dev *S0 = A.state0;
  float freqw = S0->freqw, agc = S0->agc_gain, est_insp = S0->est_insp;
  const float min_f = S0->min_freqw, max_f = S0->max_freqw;
  float fwin = 65536.0f / C.omega / 2048.0f;
  if (fwin < 8.f) fwin = 8.f;
  const float f_lo = freqw - fwin, f_hi = freqw + fwin;
  const float kk = C.kest, k1 = 1 - C.kest;
  const float freq_alpha = C.freq_alpha, freq_beta = C.freq_beta, gain_mu = C.gain_mu, omega = C.omega;
  // symbol timing at the tile's first sample, predicted from the capture's first sample (mu = 0 there) at the nominal omega: see rx_tile_tol
  float mu = 0.f, phase = 0.f;
  {
    const double ws = (double)cb * kChunk - (double)S0->mu, om = (double)C.omega;
    const double r = ws - om * __builtin_floor(ws / om);
    mu = (float)(r > 0.0 ? om - r : 0.0);
    if (!(mu >= 0.f && mu < C.omega)) mu = 0.f;
  }
  float h1pr = 0.f, h1pi = 0.f, h1cr = 0.f, h1ci = 0.f, mmA = 0.f, mmB = 0.f;      // Mueller & Müller: the previous symbol; p1·c2 and c1·p2

  // notch: pole / gain of the interval the tile is in, S in front of the tile's first sample from the pre-pass sums
  float npr = 0.f, npi = 0.f, nkk = 0.f, sr = 0.f, si = 0.f;
  rxb_notch NS; NS.pr = NS.pi = NS.k = NS.sr = NS.si = 0.f; NS.ivm = 0;
  if (NOTCH && valid) {
    NS = rxb_notch_start(A, cap, cb * kChunk);
    npr = NS.pr; npi = NS.pi; nkk = NS.k; sr = NS.sr; si = NS.si;
  }

  unsigned hacc = 0, hcnt = 0, hwarm = 0, hnwarm = 0, got = 0;
  unsigned *const hcol = cap.hstage + j;
  float mu_begin = 0.f, phase_begin = 0.f;
  int n = 0;                                     // sample of the next symbol (tile-relative)
  unsigned x0w = 0, x1w = 0;                     // the cu8 items of samples n and n + 1
  bool fetched = false;

  for (int ci = 0; ci < wave_chunks; ++ci) {
    const bool active = ci < nchunks;
    const bool body = ci >= nwarm;
    if (active && ci == nwarm) {
      // the loop state AT the body's first sample (rx_tiling.h compares it with the previous tile's at the same sample): the next symbol
      // instant is n, `over` samples into the body
      const float over = (float)(n - ci * kChunk);
      mu_begin = mu + over; phase_begin = phase - over * freqw;
      got = hcnt; hwarm = hacc; hnwarm = hcnt < 16u ? hcnt : 16u; hcnt = 0;
    }
    // sampler->update_freq(freqw), sdr.h:790: the partner sample's extra rotation e^{−j·freqw}, constant over the chunk
    const float frev = freqw * (-1.0f / 65536.0f);
    const float cf = __builtin_amdgcn_cosf(frev), sf = __builtin_amdgcn_sinf(frev);
    if (NOTCH && active) {                       // a detect point is the start of a 4096-sample block
      const unsigned long long pos = (cb + (unsigned long long)ci) * kChunk;
      if ((pos & 4095ull) == 0) {
        NS.sr = sr; NS.si = si;
        rxb_notch_block(A, cap, pos, NS);
        npr = NS.pr; npi = NS.pi; nkk = NS.k; sr = NS.sr; si = NS.si;
      }
    }
    const unsigned cnt0 = hcnt;
    float g0r = 0.f, g0i = 0.f;                  // last interpolated sample of the chunk, before derotation (|.|² feeds the AGC)
#pragma unroll 1
    for (int sb = 0; sb < kChunk / kStage; ++sb) {
      const int s0 = ci * kChunk + sb * kStage, send = s0 + kStage;
      {
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(s0 * 2);
#pragma unroll
        for (int q = 0; q < kStageLoads; ++q)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (rx_lds_ptr)(size_t)(unsigned)(unsigned long long)(lds + q * 1024), 16, src_off[q], soff, 0, 0);
#ifndef RXB_NOWAIT                                // (measurement variant: what the stage wait costs; garbage results)
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the stage has landed (single-wave workgroup: no barrier)
#endif
        asm volatile("" ::: "memory");
      }
      const char *ap = row + 2 * (n - s0);       // sample n in this lane's row
      if (active && !fetched) {
        x0w = *reinterpret_cast<const unsigned short *>(ap); x1w = *reinterpret_cast<const unsigned short *>(ap + 2);
        fetched = true;
      }
      auto symbol = [&](auto body_tag) {
        constexpr bool BODY = decltype(body_tag)::value;
        const float x0r = rxb_u8f(x0w, 0), x0i = rxb_u8f(x0w, 1), x1r = rxb_u8f(x1w, 0), x1i = rxb_u8f(x1w, 1);
        // the items two and three samples on: the next symbol's pair is (x1, r2) or (r2, r3)
        const unsigned r2 = *reinterpret_cast<const unsigned short *>(ap + 4), r3 = *reinterpret_cast<const unsigned short *>(ap + 6);
        float o0r = x0r, o0i = x0i, o1r = x1r, o1i = x1i, t0r = 0.f, t0i = 0.f, t1r = 0.f, t1i = 0.f;
        if (NOTCH) {
          rxb_notch_step(npr, npi, nkk, sr, si, x0r, x0i, t0r, t0i);
          rxb_notch_step(npr, npi, nkk, t0r, t0i, x1r, x1i, t1r, t1i);
          o0r = x0r - t0r; o0i = x0i - t0i; o1r = x1r - t1r; o1i = x1i - t1i;
        }
        // linear_sampler::interp (sdr.h:614-623), one derotation
        const float q1r = __builtin_fmaf(o1r, cf, -(o1i * sf)), q1i = __builtin_fmaf(o1r, sf, o1i * cf);
        g0r = __builtin_fmaf(mu, q1r - o0r, o0r); g0i = __builtin_fmaf(mu, q1i - o0i, o0i);
        const float prev = phase * (-1.0f / 65536.0f);
        const float ear = __builtin_amdgcn_cosf(prev) * agc, eai = __builtin_amdgcn_sinf(prev) * agc;
        const float svr = __builtin_fmaf(g0r, ear, -(g0i * eai)), svi = __builtin_fmaf(g0r, eai, g0i * ear);
        // cstln_lut<256>::lookup (sdr.h:470-483): fold into range, truncate; the decision is the two sign bits
        float Ir = svr, Qr = svi;
        if (__builtin_fmaxf(__builtin_fabsf(svr), __builtin_fabsf(svi)) > 127.0f) lut_halve(Ir, Qr);     // (the exact test is lut_halve's own)
        const int Ii = (int)Ir, Qi = (int)Qr;
        hacc = __builtin_amdgcn_alignbit(hacc, (unsigned)Ii, 31);
        hacc = __builtin_amdgcn_alignbit(hacc, (unsigned)Qi, 31);
        const float pe = rxb_phase_error(Ii, Qi);
        phase = __builtin_fmaf(pe, freq_alpha, phase);                       // sdr.h:814-815
        freqw = __builtin_fmaf(pe, freq_beta, freqw);
        freqw = __builtin_amdgcn_fmed3f(freqw, f_lo, f_hi);
        // constellation point (±53, ±53) by the sign bits; modified Mueller & Müller, sdr.h:822-840
        float c0r, c0i;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(c0r) : "s"(0x80000000u), "v"(Ii), "v"(0x42540000u));      // 53.0 with the coordinate's sign
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(c0i) : "s"(0x80000000u), "v"(Qi), "v"(0x42540000u));
        // muerr = (p0 − p2)·c1 − (c0 − c2)·p1 (dot products) = A0 − B1 − B0 + A1 with A_k = p_k·c_{k−1}, B_k = c_k·p_{k−1}: the symbol before
        // the previous one enters through two running products, not through four more history registers
        const float A0 = __builtin_fmaf(svi, h1ci, svr * h1cr), B0 = __builtin_fmaf(c0i, h1pi, c0r * h1pr);
        const float muerr = (A0 - mmB) + (mmA - B0);
        const float mucorr = __builtin_amdgcn_fmed3f(muerr * gain_mu, -0.1f, 0.1f);
        mmA = A0; mmB = B0;
        h1pr = svr; h1pi = svi; h1cr = c0r; h1ci = c0i;
        const float mu2 = (mu + mucorr) + omega;
        // the sample steps up to the next symbol instant: at least one (sdr.h:800-847: one symbol per sample step at most)
        const float kf = __builtin_amdgcn_fmed3f(__builtin_floorf(mu2), 1.0f, 1024.0f);
        const int ki = (int)kf;
        mu = mu2 - kf;
        phase = __builtin_fmaf(kf, freqw, phase);
        ++hcnt;
        if (BODY) { if ((hcnt & 15u) == 0) hcol[(unsigned long long)((hcnt >> 4) - 1) * cap.hpitch] = hacc; }
        const bool one = ki == 1;
        x0w = one ? x1w : r2; x1w = one ? r2 : r3;
        if (NOTCH) { sr = one ? t0r : t1r; si = one ? t0i : t1i; }
        if (ki > 2) {                             // (omega > 2, or a timing excursion): walk the samples in between
          for (int i = 2; i < ki; ++i) {
            const unsigned w = *reinterpret_cast<const unsigned short *>(ap + 2 * i);
            if (NOTCH) {
              const float xr = rxb_u8f(w, 0), xi = rxb_u8f(w, 1);
              float nr, ni;
              rxb_notch_step(npr, npi, nkk, sr, si, xr, xi, nr, ni);
              sr = nr; si = ni;
            }
          }
          x0w = *reinterpret_cast<const unsigned short *>(ap + 2 * ki); x1w = *reinterpret_cast<const unsigned short *>(ap + 2 * ki + 2);
        }
        n += ki; ap += 2 * ki;
      };
      if (body) { while (active && n < send) symbol(std::true_type()); }
      else { while (active && n < send) symbol(std::false_type()); }
    }
    if (active) {
      phase = fmod65536(phase);                                              // sdr.h:855
      if (hcnt != cnt0) {                                                    // the chunk had a symbol: sdr.h:867-870
        const float insp = g0r * g0r + g0i * g0i;
        est_insp = __builtin_fmaf(insp, kk, est_insp * k1);
        if (est_insp) agc = kCstlnAmp / __builtin_sqrtf(est_insp);
      }
      if (!C.allow_drift) {                                                  // sdr.h:895-898
        if (freqw < min_f || freqw > max_f) freqw = (max_f + min_f) / 2;
      }
    }
  }
  if (valid) {
    if (nchunks <= nwarm) { mu_begin = mu; phase_begin = phase; got = hcnt; hwarm = hacc; hnwarm = hcnt < 16u ? hcnt : 16u; hcnt = 0; }   // (never: a tile has a body)
    const float over_end = (float)(n - nchunks * kChunk);     // … and AT the sample behind the tile's last one
    mu += over_end; phase -= over_end * freqw;
    if (hcnt & 15u) hcol[(unsigned long long)(hcnt >> 4) * cap.hpitch] = hacc << (2 * (16 - (hcnt & 15u)));
    rx_tile_info_h th;
    th.mu_begin = mu_begin; th.phase_begin = phase_begin; th.mu_end = mu; th.phase_end = phase;
    th.count = hcnt; th.has_pre = got ? 1u : 0u; th.n_warm = hnwarm; th.warm_tail = hwarm; th.body_tail = hacc;
    cap.hinfo[j] = th;
    if (j == cap.n_tiles - 1) { cap.state_end->phase = phase; cap.state_end->freqw = freqw; }
  }
}

template <bool NOTCH>
__global__ __launch_bounds__(64) void k_rxb_tiles(rxb_args A) {
  __shared__ __attribute__((aligned(16))) char lds[64 * rxb_stage::kRowBytes];
  const rxb_cap &cap = A.caps[blockIdx.y];
  if (blockIdx.x == 0) {
    // tile 0: the reference's arithmetic from the constructed state over the first warm_chunks chunks (in front of the first detect
    // point the notch passes its input through: SURVEY A7)
    if (threadIdx.x == 0 && cap.n_tiles) {
      rx_tiled_args a;
      a.in = cap.in; a.total_chunks = cap.total_chunks; a.first_chunks = A.warm_chunks; a.tile_chunks = A.tile_chunks; a.warm_chunks = A.warm_chunks;
      a.n_tiles = cap.n_tiles; a.lanes_per_wave = 64; a.dbg = 0; a.stage_stride = 0; a.stage = nullptr; a.wstage = nullptr; a.wstride = 0;
      a.info = nullptr; a.hstage = cap.hstage; a.hpitch = cap.hpitch; a.hinfo = cap.hinfo; a.ema = cap.ema_scratch; a.ema_wave = cap.ema_scratch + 1;
      a.state = A.state0; a.state_next = cap.state_end; a.meas = nullptr; a.meas_base = 0; a.cstln = nullptr; a.C = A.C; a.T = A.T;
      rx_tile_exact<1, LSDR_IN_CU8, true>(a);
    }
    return;
  }
  const unsigned j0 = 1u + (blockIdx.x - 1u) * 64u;
  if (j0 >= cap.n_tiles) return;
  rxb_tile<NOTCH>(A, cap, j0, (int)threadIdx.x, lds);
}

__global__ __launch_bounds__(kSeamBlock) void k_rxb_seam(rxb_args A, float omega, int R, float quad, const uint8_t *relabel) {
  const rxb_cap &cap = A.caps[blockIdx.y];
  if (blockIdx.x * kSeamBlock >= cap.n_tiles) return;
  rx_seam_h_body(cap.hinfo, cap.fix, cap.n_tiles, omega, R, quad, cap.part, relabel);
}
constexpr unsigned kRxbCompactLanes = 4;      // lanes per tile in the compaction (rx_tiling.h)
__global__ __launch_bounds__(64) void k_rxb_compact(rxb_args A, int R, float quad, const uint8_t *relabel) {
  const rxb_cap &cap = A.caps[blockIdx.y];
  if (blockIdx.x * (64u / kRxbCompactLanes) >= cap.n_tiles) return;
  rx_compact_h_body<rx_state_dev, (int)kRxbCompactLanes>(cap.hstage, cap.hpitch, cap.hinfo, cap.fix, cap.part, relabel, cap.n_tiles, R, quad, cap.out_words, 0ull,
                                  cap.state_end, cap.res);
}

#endif  // LSDR_RXB_DEVICE_H
