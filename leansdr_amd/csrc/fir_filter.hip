// leansdr_amd/csrc/fir_filter.hip — decimating complex FIR for gfx950.
//
// Replaces fir_filter<cf32,float>::run + set_freq (dsp.h:219-285):
//     y[m] = Σ_{i=0}^{N-1} sc[i] · x[N + m·D − i]          (i ascending)
// with complex·complex = (a.re·b.re − a.im·b.im, a.re·b.im + a.im·b.re)
// (math.h:40-43, a = coefficient, b = sample) and x += product per tap.
//
// Design (MI355X-first; HBM-streaming kernel, no MFMA — exact f32 order matters):
//  * One workgroup = 256 lanes = one tile of M = 256·R consecutive outputs.
//    The tile's input span ((M−1)·D + N samples) is staged ONCE from HBM into
//    LDS, so each input sample is fetched from HBM once (plus the N−D overlap
//    between neighbouring tiles, ≈3.6 % at N=313/D=30, normally an L2 hit
//    because neighbouring tiles are mapped to the same XCD).
//  * Polyphase ("overlap-save, transposed") LDS layout: tile-local sample
//    t = q·D + p is stored at row p, column q:  lds[p·S + q].  Lane l working
//    on output m0+l needs, for tap u = N−i, sample t = l·D + u, i.e. row
//    (u mod D), column (u div D) + l: for a fixed tap all 64 lanes read 64
//    CONSECUTIVE 8-byte words → conflict-free ds_read_b64, and the (row, col)
//    walk is wave-uniform (SGPR arithmetic only).  S is odd so that the
//    transposing stores (consecutive lanes → consecutive rows) are
//    conflict-free for ds_write_b64 too.
//  * Coefficients are wave-uniform → scalar loads (s_load) from a small global
//    array that lives in the scalar cache; VALU instructions take them as
//    SGPR operands.
//  * Exact mode keeps the reference's operation order with FP contraction off
//    (the file is compiled with -ffp-contract=off): 8 VALU/tap for complex
//    taps.  When every shifted coefficient has a zero imaginary part
//    (current_freq == 0, the steady state of leandvb --resample) the products
//    with ±0 cannot change any accumulator bit for finite inputs, and a
//    4-VALU/tap real-coefficient kernel is used; it is bit-identical.
//  * Fused input stage: cu8→f32 (cconverter, dsp.h:40-50) or ×scale (scaler,
//    dsp.h:149-156) is applied while staging, so the converted/scaled stream
//    never exists in HBM.
//  * blockIdx → tile mapping is XCD-aware: block b runs on XCD b%8, so XCD k
//    gets the k-th contiguous eighth of the tiles.
//
// Roofline: algorithmic bytes per input sample = 8 (cf32) or 2 (cu8) read
// + 8/D written.  N/D·{4|8} VALU lane-ops and N/D LDS 8-byte reads per input
// sample.
#include "lsdr_internal.h"

namespace {

constexpr int kThreads = 256;

struct fir_args {
  const void *in;        // cf32 or cu8 samples
  float2 *out;
  const float2 *sc;      // shifted coefficients [N]   (complex kernels)
  const float *rc;       // real coefficients   [N]   (real kernel)
  unsigned N, D;
  unsigned S;            // LDS row stride in samples (odd)
  unsigned long long count;      // outputs to produce
  unsigned long long n_in;       // input samples available
  unsigned n_tiles, tiles_per_xcd;
  float in_scale;        // 0 → none
};

// Staging is split into the global load (raw bits kept in two VGPRs) and the
// conversion applied just before the LDS write.
template <int IN_FMT>
__device__ __forceinline__ float2 load_raw(const void *in, unsigned long long j) {
  if (IN_FMT == LSDR_IN_CU8) {
    const unsigned short r = reinterpret_cast<const unsigned short *>(in)[j];
    return make_float2(__uint_as_float((unsigned)r), 0.f);
  }
  return reinterpret_cast<const float2 *>(in)[j];
}

template <int IN_FMT>
__device__ __forceinline__ float2 finish_sample(float2 raw, float scale) {
  float2 v = raw;
  if (IN_FMT == LSDR_IN_CU8) {
    const unsigned r = __float_as_uint(raw.x);
    v.x = (float)((int)(r & 0xffu) - 128);  // dsp.h:46-47: int arithmetic, then int→float
    v.y = (float)((int)(r >> 8) - 128);
  }
  if (scale != 0.f) {  // scaler: complex*T = (re*k, im*k), math.h:45-48
    v.x = v.x * scale;
    v.y = v.y * scale;
  }
  return v;
}

// One tap for the R outputs of a lane.  MODE: 0 exact complex, 1 exact real-coefficient,
// 2 FMA complex, 3 FMA real.  px points at the lane's sample for output r=0.
template <int R, int MODE>
__device__ __forceinline__ void fir_tap(const float2 *__restrict__ psc, const float *__restrict__ prc,
                                        const float2 *px, float (&accr)[R], float (&acci)[R]) {
  if (MODE == 0 || MODE == 2) {
    const float2 c = *psc;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float2 x = px[r * kThreads];
      if (MODE == 0) {
        float pr = c.x * x.x - c.y * x.y;
        float pq = c.x * x.y + c.y * x.x;
        accr[r] = accr[r] + pr;
        acci[r] = acci[r] + pq;
      } else {
        accr[r] = __builtin_fmaf(c.x, x.x, accr[r]);
        accr[r] = __builtin_fmaf(-c.y, x.y, accr[r]);
        acci[r] = __builtin_fmaf(c.x, x.y, acci[r]);
        acci[r] = __builtin_fmaf(c.y, x.x, acci[r]);
      }
    }
  } else {
    const float c = *prc;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float2 x = px[r * kThreads];
      if (MODE == 1) {
        accr[r] = accr[r] + c * x.x;
        acci[r] = acci[r] + c * x.y;
      } else {
        accr[r] = __builtin_fmaf(c, x.x, accr[r]);
        acci[r] = __builtin_fmaf(c, x.y, acci[r]);
      }
    }
  }
}

// DT > 0: decimation and LDS row stride are compile-time (S = 256·R + kSpad), so a
// whole polyphase column (D taps) is one straight-line block whose LDS reads use
// immediate offsets and whose coefficients arrive by wide scalar loads.
// DT == 0: generic run-time D / S.
constexpr unsigned kSpad = 13;

template <int IN_FMT, int R, int MODE, int DT>
__global__ __launch_bounds__(kThreads) void k_fir(fir_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2 *lds = reinterpret_cast<float2 *>(smem_raw);

  // XCD-aware tile mapping (block b is dispatched to XCD b % 8).
  const unsigned b = blockIdx.x;
  const unsigned tile = (b & 7u) * a.tiles_per_xcd + (b >> 3);
  if (tile >= a.n_tiles) return;

  constexpr unsigned M = kThreads * R;
  const unsigned long long m0 = (unsigned long long)tile * M;
  const unsigned long long rem = a.count - m0;
  const unsigned mv = rem < M ? (unsigned)rem : M;  // valid outputs in this tile
  const unsigned N = a.N;
  const unsigned D = DT > 0 ? (unsigned)DT : a.D;
  const unsigned S = DT > 0 ? (M + kSpad) : a.S;
  const unsigned l = threadIdx.x;

  // ---- stage: tile-local t ∈ [0, T) ↔ global sample j = m0·D + t
  // Consecutive lanes ↔ consecutive samples: coalesced loads; the (row, col) of
  // consecutive t differ by one row → conflict-free ds_write_b64.  ALL of a
  // lane's loads are issued before the first LDS write (≈64 KB in flight per
  // workgroup): HBM latency is paid once per tile, not once per load.
  const unsigned T = (mv - 1) * D + N + 1;
  const unsigned long long j0 = m0 * D;
  if (DT > 0) {
    // N ≤ 12·D − 1 for specialised kernels (N/D + 2 ≤ kSpad) → T ≤ (M + 11)·D
    constexpr int NL = DT > 0 ? (int)(((M + kSpad - 2) * (unsigned)(DT > 0 ? DT : 1) + kThreads - 1) / kThreads) : 1;
    float2 v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const unsigned t = l + k * kThreads;
      if (t < T) v[k] = load_raw<IN_FMT>(a.in, j0 + t);
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const unsigned t = l + k * kThreads;
      if (t < T) lds[(t % D) * S + (t / D)] = finish_sample<IN_FMT>(v[k], a.in_scale);
    }
  } else {
    constexpr int NB = 8;
    for (unsigned tb = 0; tb < T; tb += NB * kThreads) {
      float2 v[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const unsigned t = tb + l + k * kThreads;
        if (t < T) v[k] = load_raw<IN_FMT>(a.in, j0 + t);
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const unsigned t = tb + l + k * kThreads;
        if (t < T) lds[(t % D) * S + (t / D)] = finish_sample<IN_FMT>(v[k], a.in_scale);
      }
    }
  }
  __syncthreads();

  // ---- taps.  Tap i ↔ u = N − i = col·D + row, visited col-major descending.
  float accr[R], acci[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { accr[r] = 0.f; acci[r] = 0.f; }

  const float2 *base = lds + l;
  // coefficient cursors (wave-uniform → scalar loads; pointer + constant offsets
  // lets the compiler merge a column's taps into wide s_load_dwordx8/x16)
  const float2 *__restrict__ psc = a.sc;
  const float *__restrict__ prc = a.rc;
  int col = (int)(N / D);
  // leading partial column: rows (N mod D) … 0 (… 1 when it is also column 0)
  {
    const int lo = col == 0 ? 1 : 0;
    const float2 *px = base + (N % D) * S + (unsigned)col;
    for (int row = (int)(N % D); row >= lo; --row, ++psc, ++prc, px -= S) fir_tap<R, MODE>(psc, prc, px, accr, acci);
    --col;
  }
  // full columns col … 1: rows D−1 … 0
  for (; col >= 1; --col) {
    const float2 *px = base + (D - 1) * S + (unsigned)col;
    if (DT > 0) {
#pragma unroll
      for (int k = 0; k < (DT > 0 ? DT : 1); ++k) fir_tap<R, MODE>(psc + k, prc + k, px - k * (int)S, accr, acci);
      psc += D; prc += D;
    } else {
#pragma unroll 4
      for (unsigned k = 0; k < D; ++k, ++psc, ++prc, px -= S) fir_tap<R, MODE>(psc, prc, px, accr, acci);
    }
  }
  // trailing column 0: rows D−1 … 1   (u = 0 is not a tap)
  if (col == 0) {
    const float2 *px = base + (D - 1) * S;
    for (int row = (int)D - 1; row >= 1; --row, ++psc, ++prc, px -= S) fir_tap<R, MODE>(psc, prc, px, accr, acci);
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned lm = l + r * kThreads;
    if (lm < mv) a.out[m0 + lm] = make_float2(accr[r], acci[r]);
  }
}

typedef void (*fir_kernel_t)(fir_args);

template <int IN_FMT, int R, int DT>
fir_kernel_t pick_mode(int mode) {
  switch (mode) {
    case 0: return k_fir<IN_FMT, R, 0, DT>;
    case 1: return k_fir<IN_FMT, R, 1, DT>;
    case 2: return k_fir<IN_FMT, R, 2, DT>;
    default: return k_fir<IN_FMT, R, 3, DT>;
  }
}

// Specialised (compile-time D) kernels exist for the decimations listed here;
// R is fixed per D by the LDS budget: S·D·8 B ≤ ~66 KB and (D−1)·S·8 < 65536
// (16-bit DS immediate offsets).
constexpr int spec_r(int D) { return D <= 8 ? 4 : (D <= 15 ? 2 : 1); }

template <int IN_FMT>
fir_kernel_t pick_spec(unsigned D, int mode, int *R_out) {
#define LSDR_FIR_SPEC(DD) case DD: *R_out = spec_r(DD); return pick_mode<IN_FMT, spec_r(DD), DD>(mode);
  switch (D) {
    LSDR_FIR_SPEC(1) LSDR_FIR_SPEC(2) LSDR_FIR_SPEC(4) LSDR_FIR_SPEC(5) LSDR_FIR_SPEC(8)
    LSDR_FIR_SPEC(10) LSDR_FIR_SPEC(16) LSDR_FIR_SPEC(30)
    default: return nullptr;
  }
#undef LSDR_FIR_SPEC
}

template <int IN_FMT>
fir_kernel_t pick_generic(int R, int mode) {
  switch (R) {
    case 1: return pick_mode<IN_FMT, 1, 0>(mode);
    case 2: return pick_mode<IN_FMT, 2, 0>(mode);
    default: return pick_mode<IN_FMT, 4, 0>(mode);
  }
}

}  // namespace

struct lsdr_fir_filter {
  lsdr_ctx *ctx;
  lsdr_fir_filter_cfg cfg;
  std::vector<float> coeffs;        // prototype (host)
  std::vector<lsdr_cf32> shifted;   // host copy of shifted_coeffs
  float2 *d_sc;                     // device: shifted coefficients
  float *d_rc;                      // device: real parts (valid when all imag == 0)
  bool all_real;
  float current_freq;
  int R;                            // outputs per lane
  unsigned S;                       // LDS row stride
  size_t lds_bytes;
  int force_complex;                // test hook (env LSDR_FIR_FORCE_COMPLEX)
  bool spec;                        // compile-time-D kernel in use
};

static int fir_upload(lsdr_fir_filter *f) {
  const unsigned N = f->cfg.ncoeffs;
  f->all_real = true;
  std::vector<float> rc(N);
  for (unsigned i = 0; i < N; ++i) {
    if (f->shifted[i].im != 0.0f) f->all_real = false;
    rc[i] = f->shifted[i].re;
  }
  lsdr_ctx *c = f->ctx;
  // The previous coefficient set may still be in use by queued launches.
  LSDR_HIP(hipStreamSynchronize(c->stream));
  LSDR_HIP(hipMemcpyAsync(f->d_sc, f->shifted.data(), N * sizeof(float2), hipMemcpyHostToDevice, c->stream));
  LSDR_HIP(hipMemcpyAsync(f->d_rc, rc.data(), N * sizeof(float), hipMemcpyHostToDevice, c->stream));
  LSDR_HIP(hipStreamSynchronize(c->stream));
  return LSDR_OK;
}

extern "C" {

int lsdr_fir_filter_create(lsdr_ctx *c, const lsdr_fir_filter_cfg *cfg, lsdr_fir_filter **out) {
  LSDR_ARG(c && cfg && out);
  LSDR_ARG(cfg->ncoeffs >= 1 && cfg->coeffs_host && cfg->decim >= 1);
  LSDR_ARG(cfg->in_format == LSDR_IN_CF32 || cfg->in_format == LSDR_IN_CU8);
  LSDR_ARG(cfg->arith == LSDR_FIR_EXACT || cfg->arith == LSDR_FIR_FMA);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_fir_filter *f = new lsdr_fir_filter();
  f->ctx = c;
  f->cfg = *cfg;
  f->coeffs.assign(cfg->coeffs_host, cfg->coeffs_host + cfg->ncoeffs);
  f->cfg.coeffs_host = f->coeffs.data();
  f->shifted.resize(cfg->ncoeffs);
  const char *fc = getenv("LSDR_FIR_FORCE_COMPLEX");
  f->force_complex = fc && atoi(fc);

  // Tile geometry: R outputs per lane so that the LDS tile stays ≤ 64 KiB
  // (two workgroups per CU overlap each other's staging and tap phases).
  const unsigned N = cfg->ncoeffs, D = cfg->decim;
  const char *fr = getenv("LSDR_FIR_R");
  const char *fg = getenv("LSDR_FIR_GENERIC");   // test hook: force the run-time-D kernel
  int R = fr ? atoi(fr) : 0;
  auto lds_for = [&](int r, unsigned *S_out) {
    unsigned M = kThreads * r;
    unsigned Q = M + N / D + 2;   // columns: q ≤ (M-1) + N/D (+1 for t = T-1 rounding)
    unsigned S = Q | 1;           // odd row stride
    *S_out = S;
    return (size_t)D * S * sizeof(float2);
  };
  unsigned S;
  size_t bytes;
  int Rs = 0;
  f->spec = false;
  if (!(fg && atoi(fg)) && N / D + 2 <= kSpad && pick_spec<LSDR_IN_CF32>(D, 0, &Rs) != nullptr) {
    f->spec = true;
    R = Rs;
    S = kThreads * R + kSpad;
    bytes = (size_t)D * S * sizeof(float2);
  } else {
    if (R != 1 && R != 2 && R != 4) {
      R = 4;
      while (R > 1 && lds_for(R, &S) > 64 * 1024) R >>= 1;
    }
    bytes = lds_for(R, &S);
  }
  f->R = R;
  f->S = S;
  f->lds_bytes = bytes;
  LSDR_HIP(hipMalloc((void **)&f->d_sc, N * sizeof(float2)));
  LSDR_HIP(hipMalloc((void **)&f->d_rc, N * sizeof(float)));
  *out = f;
  return lsdr_fir_filter_set_freq(f, 0.0f);  // fir_filter ctor ends with set_freq(0), dsp.h:230
}

void lsdr_fir_filter_destroy(lsdr_fir_filter *f) {
  if (!f) return;
  (void)hipStreamSynchronize(f->ctx->stream);
  (void)hipFree(f->d_sc);
  (void)hipFree(f->d_rc);
  delete f;
}

int lsdr_fir_filter_set_freq(lsdr_fir_filter *f, float freq) {
  LSDR_ARG(f);
  lsdr::fir_shift_coeffs(f->cfg.ncoeffs, f->coeffs.data(), freq, f->shifted.data());
  f->current_freq = freq;
  return fir_upload(f);
}

int lsdr_fir_filter_track(lsdr_fir_filter *f, float freq_tap, float tap_multiplier, float freq_tol, int *shifted) {
  LSDR_ARG(f);
  // dsp.h:237-238: new_freq in float; fabs() of the float difference compared with freq_tol.
  float new_freq = freq_tap * tap_multiplier;
  int did = 0;
  if (fabs(f->current_freq - new_freq) > freq_tol) {
    int rc = lsdr_fir_filter_set_freq(f, new_freq);
    if (rc) return rc;
    did = 1;
  }
  if (shifted) *shifted = did;
  return LSDR_OK;
}

float lsdr_fir_filter_current_freq(const lsdr_fir_filter *f) { return f ? f->current_freq : 0.f; }

int lsdr_fir_filter_get_shifted_coeffs(const lsdr_fir_filter *f, lsdr_cf32 *out) {
  LSDR_ARG(f && out);
  memcpy(out, f->shifted.data(), f->shifted.size() * sizeof(lsdr_cf32));
  return LSDR_OK;
}

int lsdr_fir_filter_run(lsdr_fir_filter *f, const void *in, size_t n_in, lsdr_cf32 *out, size_t cap_out,
                        size_t *consumed, size_t *produced) {
  LSDR_ARG(f && consumed && produced);
  *consumed = 0;
  *produced = 0;
  const unsigned N = f->cfg.ncoeffs, D = f->cfg.decim;
  if (n_in < N) return LSDR_OK;  // dsp.h:234
  size_t count = (n_in - N) / D;
  if (count > cap_out) count = cap_out;
  if (!count) return LSDR_OK;
  LSDR_ARG(in && out);

  fir_args a;
  a.in = in;
  a.out = (float2 *)out;
  a.sc = f->d_sc;
  a.rc = f->d_rc;
  a.N = N; a.D = D; a.S = f->S;
  a.count = count;
  a.n_in = n_in;
  const unsigned M = kThreads * f->R;
  size_t n_tiles = (count + M - 1) / M;
  LSDR_ARG(n_tiles < (1ull << 31));
  a.n_tiles = (unsigned)n_tiles;
  a.tiles_per_xcd = (unsigned)((n_tiles + 7) / 8);
  a.in_scale = f->cfg.in_scale;

  const bool real_path = f->all_real && !f->force_complex;
  int mode = (f->cfg.arith == LSDR_FIR_FMA ? 2 : 0) + (real_path ? 1 : 0);
  int Rs = 0;
  fir_kernel_t k;
  if (f->spec)
    k = f->cfg.in_format == LSDR_IN_CU8 ? pick_spec<LSDR_IN_CU8>(D, mode, &Rs) : pick_spec<LSDR_IN_CF32>(D, mode, &Rs);
  else
    k = f->cfg.in_format == LSDR_IN_CU8 ? pick_generic<LSDR_IN_CU8>(f->R, mode) : pick_generic<LSDR_IN_CF32>(f->R, mode);
  if (f->lds_bytes > 64 * 1024)
    LSDR_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f->lds_bytes));
  hipLaunchKernelGGL(k, dim3(a.tiles_per_xcd * 8), dim3(kThreads), f->lds_bytes, f->ctx->stream, a);
  LSDR_HIP(hipGetLastError());
  *produced = count;
  *consumed = count * D;
  return LSDR_OK;
}

}  // extern "C"
