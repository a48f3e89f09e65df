// leansdr_amd/csrc/chan.hip — the channel simulator of leanchansim (leanchansim.cc:34-190) on gfx950, the last piece of the
// generator side: with it a noisy DVB-S capture is synthesised where the receiver consumes it.
//
//   wgn_c<f32>                   dsp.h:164-190        polar-method Gaussian noise on glibc's drand48() and logf()
//   adder<cf32>                  dsp.h:118-138
//   drifter<float>               leanchansim.cc:34-88 LO drift: three sinusoidal FM components, 16-bit phase accumulator
//   cconverter<f32,0,u8,128,1,1> dsp.h:33-54          x86 float → int32 → u8 truncation
//
// wgn_c is bit-exact with the reference on the same seed:
//   * drand48 is the 48-bit LCG X' = a·X + c; draw number k is reached in O(log k) with the powers (a^(2^b), c_b) of the
//     affine map, so every thread starts anywhere in the stream;
//   * the rejection step (keep a pair when 0 < x²+y² < 1) makes the position of output i data dependent: pass 1 counts the
//     accepted pairs of every block, a one-block scan turns counts into offsets, pass 2 regenerates the candidates and
//     writes output i where the reference would; the state after the pair that produced the last output stays on the device;
//   * logf is glibc 2.35's (table + degree-3 polynomial in double), restated — the oracle's copy is compared with libm over
//     the whole domain (0,1) by tests/test_oracle_chan.py; sqrtf and the divisions are IEEE.
// drifter: `phase` is a local of run() (it restarts at 0 on every call, so the reference's output depends on its 4096-sample
// pipe size); the ABI takes that chunk length, a lane walks one chunk with the reference's float/double/integer conversions
// spelled out (x86 "integer indefinite" on overflow/NaN), and a parallel pass applies the rotation.
#include "lsdr_internal.h"

namespace {

constexpr unsigned long long kLcgA = 0x5DEECE66DULL, kLcgC = 0xBULL, kLcgMask = 0xFFFFFFFFFFFFULL;
constexpr int kWgnBlock = 256, kWgnPairs = 16;                    // candidate pairs per thread
constexpr unsigned kWgnPerBlock = kWgnBlock * kWgnPairs;

struct lcg_pow { unsigned long long a[48], c[48]; };               // X_{k+2^b} = a[b]·X_k + c[b]  (mod 2^48)
struct logf_tab { double invc[16], logc[16]; };

__device__ __forceinline__ unsigned long long lcg_jump(const lcg_pow &p, unsigned long long x, unsigned long long k) {
  for (int b = 0; b < 48 && (k >> b); ++b)
    if ((k >> b) & 1) x = (p.a[b] * x + p.c[b]) & kLcgMask;
  return x;
}
__device__ __forceinline__ double lcg_next(unsigned long long &x) {
  x = (x * kLcgA + kLcgC) & kLcgMask;
  return (double)x * 0x1p-48;
}
// glibc 2.35 logf for positive normal x (sysdeps/ieee754/flt-32/e_logf.c)
__device__ __forceinline__ float glibc_logf(const logf_tab &t, float x) {
  const unsigned ix = __float_as_uint(x);
  if (ix == 0x3f800000u) return 0.f;
  const unsigned tmp = ix - 0x3f330000u;
  const int i = (tmp >> 19) % 16, k = (int)tmp >> 23;
  const unsigned iz = ix - (tmp & (0x1ffu << 23));
  const double z = (double)__uint_as_float(iz), r = z * t.invc[i] - 1, y0 = t.logc[i] + (double)k * 0x1.62e42fefa39efp-1, r2 = r * r;
  double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
  y = -0x1.00ea348b88334p-2 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

// Candidate pairs [first, first+count) of the stream that starts at state *x0; thread t of block b owns pairs
// (b·256 + t)·kWgnPairs … .  Pass 1: accepted pairs per block.
__global__ __launch_bounds__(kWgnBlock) void k_wgn_count(const unsigned long long *x0, const lcg_pow *pw, unsigned long long npairs,
                                                         unsigned *block_count) {
  __shared__ unsigned red[kWgnBlock / 64];
  const unsigned long long first = ((unsigned long long)blockIdx.x * kWgnBlock + threadIdx.x) * kWgnPairs;
  unsigned cnt = 0;
  if (first < npairs) {
    unsigned long long x = lcg_jump(*pw, *x0, 2 * first);
    const unsigned m = (unsigned)((npairs - first) < kWgnPairs ? (npairs - first) : kWgnPairs);
    for (unsigned j = 0; j < m; ++j) {
      const float u = (float)(2 * lcg_next(x) - 1), v = (float)(2 * lcg_next(x) - 1);
      const float r2 = u * u + v * v;
      cnt += !(r2 == 0 || r2 >= 1);
    }
  }
  for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// Exclusive scan of the block counts (one block; totals in block_off[nblocks]).
__global__ __launch_bounds__(1024) void k_wgn_scan(const unsigned *block_count, unsigned nblocks, unsigned long long *block_off) {
  __shared__ unsigned long long part[1024];
  const unsigned per = (nblocks + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < nblocks ? lo + per : nblocks;
  unsigned long long s = 0;
  for (unsigned i = lo; i < hi; ++i) s += block_count[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (unsigned o = 1; o < 1024; o <<= 1) {
    const unsigned long long v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned long long run = part[threadIdx.x] - s;
  for (unsigned i = lo; i < hi; ++i) { block_off[i] = run; run += block_count[i]; }
  if (threadIdx.x == 1023) block_off[nblocks] = part[1023];
}
// Pass 2: regenerate, place output i at out[i] for i < n; the thread holding output n−1 (or, if the candidates run out first,
// the last candidate) records the stream state for the next call.  res[0] = outputs written, res[1] = next state.
__global__ __launch_bounds__(kWgnBlock) void k_wgn_emit(const unsigned long long *x0, const lcg_pow *pw, const logf_tab *lt,
                                                        unsigned long long npairs, const unsigned long long *block_off,
                                                        unsigned nblocks, float stddev, const float2 *add, float2 *out,
                                                        unsigned long long n, unsigned long long *res) {
  __shared__ unsigned wsum[kWgnBlock / 64];
  __shared__ logf_tab t;
  if (threadIdx.x < 16) { t.invc[threadIdx.x] = lt->invc[threadIdx.x]; t.logc[threadIdx.x] = lt->logc[threadIdx.x]; }
  const unsigned long long base = block_off[blockIdx.x];
  if (base >= n) return;   // whole block lies past the last requested output (uniform)
  const unsigned long long first = ((unsigned long long)blockIdx.x * kWgnBlock + threadIdx.x) * kWgnPairs;
  float us[kWgnPairs], vs[kWgnPairs], rs[kWgnPairs];
  unsigned mask = 0, m = 0;
  unsigned long long x = 0;
  if (first < npairs) {
    x = lcg_jump(*pw, *x0, 2 * first);
    m = (unsigned)((npairs - first) < kWgnPairs ? (npairs - first) : kWgnPairs);
  }
  unsigned long long xr = x;   // replayed below to know the state after each accepted pair
#pragma unroll
  for (unsigned j = 0; j < kWgnPairs; ++j) {
    if (j < m) {
      us[j] = (float)(2 * lcg_next(x) - 1);
      vs[j] = (float)(2 * lcg_next(x) - 1);
      rs[j] = us[j] * us[j] + vs[j] * vs[j];
      if (!(rs[j] == 0 || rs[j] >= 1)) mask |= 1u << j;
    }
  }
  const unsigned cnt = __popc(mask);
  unsigned incl = cnt;   // inclusive scan over the wave, then over the 4 waves
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(incl, o);
    if ((int)(threadIdx.x & 63) >= o) incl += v;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned woff = 0;
  for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) woff += wsum[w];
  unsigned long long idx = base + woff + incl - cnt;
#pragma unroll
  for (unsigned j = 0; j < kWgnPairs; ++j) {
    if (j < m) {
      xr = (xr * kLcgA + kLcgC) & kLcgMask;
      xr = (xr * kLcgA + kLcgC) & kLcgMask;
      if ((mask >> j) & 1) {
        if (idx < n) {
          const float k = __builtin_sqrtf(-glibc_logf(t, rs[j]) / rs[j]) * stddev;
          float2 o = make_float2(k * us[j], k * vs[j]);
          if (add) { const float2 a = add[idx]; o = make_float2(a.x + o.x, a.y + o.y); }
          out[idx] = o;
          if (idx == n - 1) { res[0] = n; res[1] = xr; }
        }
        ++idx;
      }
    }
  }
  // candidates exhausted before n outputs: the very last candidate pair's owner reports what there is
  if (first < npairs && first + m == npairs && block_off[nblocks] < n) { res[0] = block_off[nblocks]; res[1] = xr; }
}

__global__ __launch_bounds__(256) void k_add(const float2 *a, const float2 *b, unsigned long long n, float2 *out) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    out[i] = make_float2(a[i].x + b[i].x, a[i].y + b[i].y);
}

// cvttss2si / cvttsd2si: the "integer indefinite" value when the source is NaN or out of range
__device__ __forceinline__ int x86_f2i(float v) { return (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000; }
__device__ __forceinline__ long long x86_d2l(double v) {
  return (v >= -9223372036854775808.0 && v < 9223372036854775808.0) ? (long long)v : (long long)0x8000000000000000ULL;
}

__global__ __launch_bounds__(256) void k_cconv_f32_u8(const float2 *in, unsigned long long n, uchar2 *out) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float2 v = in[i];
    out[i] = make_uchar2((unsigned char)x86_f2i(128 + (v.x - 0.f) * 1 / 1), (unsigned char)x86_f2i(128 + (v.y - 0.f) * 1 / 1));
  }
}

__global__ __launch_bounds__(256) void k_cconv_f32_s16(const float2 *in, unsigned long long n, short2 *out) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float2 v = in[i];
    out[i] = make_short2((short)x86_f2i(0 + (v.x - 0.f) * 32768 / 1), (short)x86_f2i(0 + (v.y - 0.f) * 32768 / 1));
  }
}

struct drift_comp { float amp, freq; long long a; int active; };
struct drift_args { drift_comp c[3]; long long step[3]; };   // step: per-sample increment where the closed form is valid

// One lane per run()-chunk: the phase sequence of that chunk (leanchansim.cc:61-75).  a_start: a[i] at the first sample of
// the call; chunk-start values are a_start + s·step (closed form checked by the host) or come from a_chunk (pre-pass).
__global__ __launch_bounds__(64) void k_drift_phase(drift_args g, const long long *a_chunk, const float2 *lut, unsigned long long n,
                                                    unsigned chunk, unsigned short *ph) {
  const unsigned long long c = (unsigned long long)blockIdx.x * 64 + threadIdx.x, s0 = c * chunk;
  if (s0 >= n) return;
  const unsigned len = (unsigned)((n - s0) < chunk ? (n - s0) : chunk);
  long long a[3];
  double d[3];
  for (int i = 0; i < 3; ++i) {
    a[i] = a_chunk ? a_chunk[c * 3 + i] : g.c[i].a + (long long)s0 * g.step[i];
    d[i] = g.c[i].freq * 4294967296.0;
  }
  short phase = 0;
  for (unsigned s = 0; s < len; ++s) {
    float f = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (!g.c[i].active) continue;
      f += g.c[i].amp * lut[(unsigned short)(a[i] >> 16)].y;
      a[i] = x86_d2l((double)a[i] + d[i]);
    }
    phase = (short)x86_f2i((float)phase + f * 65536);
    ph[s0 + s] = (unsigned short)phase;
  }
}
// Pre-pass for components outside the closed form: one lane per component walks the whole call.
__global__ void k_drift_walk(drift_args g, unsigned long long n, unsigned chunk, long long *a_chunk, long long *a_end) {
  const int i = threadIdx.x;
  if (i >= 3) return;
  long long a = g.c[i].a;
  const double d = g.c[i].freq * 4294967296.0;
  for (unsigned long long s = 0; s < n; ++s) {
    if (s % chunk == 0) a_chunk[(s / chunk) * 3 + i] = a;
    if (g.c[i].active) a = x86_d2l((double)a + d);
  }
  a_end[i] = a;
}
__global__ __launch_bounds__(256) void k_drift_apply(const float2 *in, unsigned long long n, const unsigned short *ph, const float2 *lut,
                                                     float2 *out) {
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float2 r = lut[ph ? ph[i] : 0], v = in[i];
    out[i] = make_float2(v.x * r.x - v.y * r.y, v.x * r.y + v.y * r.x);
  }
}

static unsigned grid_for(lsdr_ctx *c, unsigned long long n, unsigned per = 256) {
  unsigned long long b = (n + per - 1) / per, cap = (unsigned long long)c->num_cu * 16;
  return (unsigned)(b < cap ? (b ? b : 1) : cap);
}

}  // namespace

struct lsdr_wgn {
  lsdr_ctx *ctx;
  unsigned long long *d_state;   // [0] current X, [1..2] result slots of k_wgn_emit
  lcg_pow *d_pow;
  logf_tab *d_logf;
  unsigned *d_count;
  unsigned long long *d_off;
  size_t blocks_cap;
};
struct lsdr_drifter {
  lsdr_ctx *ctx;
  drift_comp c[3];
  float2 *d_lut;
  unsigned short *d_ph;
  long long *d_achunk;
  size_t ph_cap, chunk_cap;
};

extern "C" {

int lsdr_wgn_create(lsdr_ctx *c, int seeded, long seed, lsdr_wgn **out) {
  LSDR_ARG(c && out);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_wgn *w = new lsdr_wgn();
  w->ctx = c;
  lcg_pow p;
  p.a[0] = kLcgA; p.c[0] = kLcgC;
  for (int b = 1; b < 48; ++b) {   // (a,c)∘(a,c): X → a·(a·X + c) + c
    p.a[b] = (p.a[b - 1] * p.a[b - 1]) & kLcgMask;
    p.c[b] = (p.a[b - 1] * p.c[b - 1] + p.c[b - 1]) & kLcgMask;
  }
  // glibc 2.35 sysdeps/ieee754/flt-32/e_logf_data.c (N = 16): {1/c, log c} for the 16 sub-intervals of [0.7, 1.4)
  static const double tab[16][2] = {
      {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
      {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
      {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
      {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
      {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
      {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
      {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
      {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
  };
  logf_tab lt;
  for (int i = 0; i < 16; ++i) { lt.invc[i] = tab[i][0]; lt.logc[i] = tab[i][1]; }
  // glibc keeps the drand48 state in zeroed static storage: an unseeded process starts from X = 0; srand48(s): X = s<<16 | 0x330E
  const unsigned long long x0[3] = {seeded ? ((((unsigned long long)(unsigned)seed) << 16) | 0x330E) : 0ULL, 0, 0};
  LSDR_HIP(hipMalloc((void **)&w->d_state, sizeof(x0)));
  LSDR_HIP(hipMalloc((void **)&w->d_pow, sizeof(p)));
  LSDR_HIP(hipMalloc((void **)&w->d_logf, sizeof(lt)));
  LSDR_HIP(hipMemcpy(w->d_state, x0, sizeof(x0), hipMemcpyHostToDevice));
  LSDR_HIP(hipMemcpy(w->d_pow, &p, sizeof(p), hipMemcpyHostToDevice));
  LSDR_HIP(hipMemcpy(w->d_logf, &lt, sizeof(lt), hipMemcpyHostToDevice));
  *out = w;
  return LSDR_OK;
}
void lsdr_wgn_destroy(lsdr_wgn *w) {
  if (!w) return;
  (void)hipStreamSynchronize(w->ctx->stream);
  (void)hipFree(w->d_state); (void)hipFree(w->d_pow); (void)hipFree(w->d_logf); (void)hipFree(w->d_count); (void)hipFree(w->d_off);
  delete w;
}
int lsdr_wgn_get_state(lsdr_wgn *w, unsigned long long *x) {
  LSDR_ARG(w && x);
  LSDR_HIP(hipMemcpyAsync(x, w->d_state, 8, hipMemcpyDeviceToHost, w->ctx->stream));
  LSDR_HIP(hipStreamSynchronize(w->ctx->stream));
  return LSDR_OK;
}
int lsdr_wgn_set_state(lsdr_wgn *w, unsigned long long x) {
  LSDR_ARG(w);
  x &= kLcgMask;
  LSDR_HIP(hipMemcpyAsync(w->d_state, &x, 8, hipMemcpyHostToDevice, w->ctx->stream));
  LSDR_HIP(hipStreamSynchronize(w->ctx->stream));
  return LSDR_OK;
}
// n samples of wgn_c<f32>::run (dsp.h:169-186) into out; with add != NULL, out = add + noise (the adder of leanchansim.cc:151 fused).
int lsdr_wgn_run(lsdr_wgn *w, float stddev, const lsdr_cf32 *add, lsdr_cf32 *out, size_t n) {
  LSDR_ARG(w && (n == 0 || out));
  lsdr_ctx *c = w->ctx;
  size_t done = 0;
  while (done < n) {
    const size_t want = n - done;
    // candidates: acceptance is π/4; 6 σ of margin, the loop covers the rest
    const double sd = sqrt((double)want * 0.17);
    unsigned long long npairs = (unsigned long long)((double)want * 1.2732395447351628 + 6 * sd * 1.2732395447351628 + 64);
    const unsigned long long maxpairs = 1ULL << 31;
    if (npairs > maxpairs) npairs = maxpairs;
    const unsigned nblocks = (unsigned)((npairs + kWgnPerBlock - 1) / kWgnPerBlock);
    if (nblocks + 1 > w->blocks_cap) {
      LSDR_HIP(hipStreamSynchronize(c->stream));
      (void)hipFree(w->d_count); (void)hipFree(w->d_off);
      w->d_count = nullptr; w->d_off = nullptr;
      LSDR_HIP(hipMalloc((void **)&w->d_count, (size_t)(nblocks + 1) * sizeof(unsigned)));
      LSDR_HIP(hipMalloc((void **)&w->d_off, (size_t)(nblocks + 1) * sizeof(unsigned long long)));
      w->blocks_cap = nblocks + 1;
    }
    hipLaunchKernelGGL(k_wgn_count, dim3(nblocks), dim3(kWgnBlock), 0, c->stream, (const unsigned long long *)w->d_state, (const lcg_pow *)w->d_pow,
                       npairs, w->d_count);
    hipLaunchKernelGGL(k_wgn_scan, dim3(1), dim3(1024), 0, c->stream, (const unsigned *)w->d_count, nblocks, w->d_off);
    hipLaunchKernelGGL(k_wgn_emit, dim3(nblocks), dim3(kWgnBlock), 0, c->stream, (const unsigned long long *)w->d_state, (const lcg_pow *)w->d_pow,
                       (const logf_tab *)w->d_logf, npairs, (const unsigned long long *)w->d_off, nblocks, stddev,
                       (const float2 *)(add ? add + done : nullptr), (float2 *)(out + done), (unsigned long long)want, w->d_state + 1);
    LSDR_HIP(hipGetLastError());
    unsigned long long res[2];
    LSDR_HIP(hipMemcpyAsync(res, w->d_state + 1, sizeof(res), hipMemcpyDeviceToHost, c->stream));
    LSDR_HIP(hipMemcpyAsync(w->d_state, w->d_state + 2, 8, hipMemcpyDeviceToDevice, c->stream));
    LSDR_HIP(hipStreamSynchronize(c->stream));
    done += (size_t)res[0];
  }
  return LSDR_OK;
}

int lsdr_adder_run(lsdr_ctx *c, const lsdr_cf32 *a, const lsdr_cf32 *b, size_t n, lsdr_cf32 *out) {
  LSDR_ARG(c && (n == 0 || (a && b && out)));
  if (!n) return LSDR_OK;
  hipLaunchKernelGGL(k_add, dim3(grid_for(c, n)), dim3(256), 0, c->stream, (const float2 *)a, (const float2 *)b, (unsigned long long)n, (float2 *)out);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}
int lsdr_cconverter_f32_u8_run(lsdr_ctx *c, const lsdr_cf32 *in, size_t n, lsdr_cu8 *out) {
  LSDR_ARG(c && (n == 0 || (in && out)));
  if (!n) return LSDR_OK;
  hipLaunchKernelGGL(k_cconv_f32_u8, dim3(grid_for(c, n)), dim3(256), 0, c->stream, (const float2 *)in, (unsigned long long)n, (uchar2 *)out);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

int lsdr_cconverter_f32_s16_run(lsdr_ctx *c, const lsdr_cf32 *in, size_t n, int16_t *out) {
  LSDR_ARG(c && (n == 0 || (in && out)));
  if (!n) return LSDR_OK;
  hipLaunchKernelGGL(k_cconv_f32_s16, dim3(grid_for(c, n)), dim3(256), 0, c->stream, (const float2 *)in, (unsigned long long)n, (short2 *)out);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

int lsdr_drifter_create(lsdr_ctx *c, lsdr_drifter **out) {
  LSDR_ARG(c && out);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_drifter *d = new lsdr_drifter();
  d->ctx = c;
  std::vector<float2> lut(65536);
  for (int i = 0; i < 65536; ++i) {   // host libm, like the reference (leanchansim.cc:42-46)
    float a = 2 * M_PI * i / 65536;
    lut[i].x = cosf(a);
    lut[i].y = sinf(a);
  }
  LSDR_HIP(hipMalloc((void **)&d->d_lut, lut.size() * sizeof(float2)));
  LSDR_HIP(hipMemcpy(d->d_lut, lut.data(), lut.size() * sizeof(float2), hipMemcpyHostToDevice));
  *out = d;
  return LSDR_OK;
}
void lsdr_drifter_destroy(lsdr_drifter *d) {
  if (!d) return;
  (void)hipStreamSynchronize(d->ctx->stream);
  (void)hipFree(d->d_lut); (void)hipFree(d->d_ph); (void)hipFree(d->d_achunk);
  delete d;
}
int lsdr_drifter_set_component(lsdr_drifter *d, int i, float amp, float freq) {
  LSDR_ARG(d && i >= 0 && i < 3);
  d->c[i].amp = amp; d->c[i].freq = freq;
  return LSDR_OK;
}
int lsdr_drifter_get_phases(lsdr_drifter *d, long long a[3]) {
  LSDR_ARG(d && a);
  for (int i = 0; i < 3; ++i) a[i] = d->c[i].a;
  return LSDR_OK;
}
int lsdr_drifter_set_phases(lsdr_drifter *d, const long long a[3]) {
  LSDR_ARG(d && a);
  for (int i = 0; i < 3; ++i) d->c[i].a = a[i];
  return LSDR_OK;
}
// n samples as consecutive run() calls of `chunk` samples each (0 = one call); leanchansim's pipes hold 4096.
int lsdr_drifter_run(lsdr_drifter *d, const lsdr_cf32 *in, size_t n, lsdr_cf32 *out, size_t chunk) {
  LSDR_ARG(d && (n == 0 || (in && out)));
  if (!n) return LSDR_OK;
  lsdr_ctx *c = d->ctx;
  if (!chunk || chunk > n) chunk = n;
  LSDR_ARG(chunk <= 0xffffffffu);
  drift_args g;
  bool any = false, walk = false;
  for (int i = 0; i < 3; ++i) {
    g.c[i] = d->c[i];
    // amp = ±0 leaves f untouched (the table is finite) and freq = 0 leaves a untouched: such a component is skipped
    g.c[i].active = !(d->c[i].amp == 0 && d->c[i].freq == 0 && d->c[i].a > -(1LL << 52) && d->c[i].a < (1LL << 52));
    g.step[i] = 0;
    if (!g.c[i].active) continue;
    any = true;
    // closed form a_s = a_0 + s·⌊δ⌋: the reference computes (long)((double)a + δ) per sample; with a ≥ 0, δ ≥ 0 and all sums
    // below 2^52 the double sum is a + ⌊δ⌋ + frac(δ) rounded to a multiple of ulp ≤ 1/2, which truncates back to a + ⌊δ⌋
    // unless frac(δ) is within ulp/2 of 1
    // (the conversion truncates toward zero, so a ≤ 0 with δ ≤ 0 is the mirror image: a_s = a_0 − s·⌊|δ|⌋)
    const double delta = d->c[i].freq * 4294967296.0;
    const bool neg = delta < 0 || (delta == 0 && d->c[i].a < 0);
    const double ad = fabs(delta), aa = fabs((double)d->c[i].a);
    bool ok = (neg ? d->c[i].a <= 0 : d->c[i].a >= 0) && ad < 0x1p51 && ad == ad;
    if (ok) {
      const double fl = floor(ad), frac = ad - fl, end = aa + ((double)n + 1) * (fl + 1);
      ok = end < 0x1p52;
      if (ok) {
        int e = 0;
        (void)frexp(end, &e);                                   // end < 2^e
        const double ulp = ldexp(1.0, (e < 1 ? 1 : e) - 53);    // ulp of the largest sum of this call
        ok = frac + ulp < 1.0;
      }
      if (ok) g.step[i] = neg ? -(long long)fl : (long long)fl;
    }
    if (!ok) walk = true;
  }
  const size_t nchunks = (n + chunk - 1) / chunk;
  const unsigned short *ph = nullptr;
  if (any) {
    if (n > d->ph_cap) {
      LSDR_HIP(hipStreamSynchronize(c->stream));
      (void)hipFree(d->d_ph);
      d->d_ph = nullptr;
      LSDR_HIP(hipMalloc((void **)&d->d_ph, n * sizeof(unsigned short)));
      d->ph_cap = n;
    }
    const long long *a_chunk = nullptr;
    long long a_end[3];
    if (walk) {
      if (nchunks + 1 > d->chunk_cap) {
        LSDR_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(d->d_achunk);
        d->d_achunk = nullptr;
        LSDR_HIP(hipMalloc((void **)&d->d_achunk, (nchunks + 1) * 3 * sizeof(long long)));
        d->chunk_cap = nchunks + 1;
      }
      hipLaunchKernelGGL(k_drift_walk, dim3(1), dim3(64), 0, c->stream, g, (unsigned long long)n, (unsigned)chunk, d->d_achunk,
                         d->d_achunk + nchunks * 3);
      a_chunk = d->d_achunk;
    }
    hipLaunchKernelGGL(k_drift_phase, dim3((unsigned)((nchunks + 63) / 64)), dim3(64), 0, c->stream, g, a_chunk, (const float2 *)d->d_lut,
                       (unsigned long long)n, (unsigned)chunk, d->d_ph);
    ph = d->d_ph;
    if (walk) {
      LSDR_HIP(hipMemcpyAsync(a_end, d->d_achunk + nchunks * 3, sizeof(a_end), hipMemcpyDeviceToHost, c->stream));
      LSDR_HIP(hipStreamSynchronize(c->stream));
      for (int i = 0; i < 3; ++i) d->c[i].a = a_end[i];
    } else {
      for (int i = 0; i < 3; ++i) d->c[i].a += (long long)n * g.step[i];
    }
  }
  hipLaunchKernelGGL(k_drift_apply, dim3(grid_for(c, n)), dim3(256), 0, c->stream, (const float2 *)in, (unsigned long long)n, ph,
                     (const float2 *)d->d_lut, (float2 *)out);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

}  // extern "C"
