// leansdr_amd/csrc/notch_detect.h — the device side of auto_notch::detect() (sdr.h:76-118), shared by the blocks that need it:
// auto_notch's scan mode and the fused notch_fir (notch.hip), the capture-batch receiver (cstln_receiver.hip).  Included INSIDE the
// includer's anonymous namespace.  cfft_engine's reverse transform (dsp.h:78-110) of one 4096-sample block as two independent
// half transforms plus the last radix-2 stage folded into the peak search: the same butterflies, each evaluated once with the
// same expression — bit-identical to k_cfft and therefore to the reference's FFT (tests/test_gpu_notch.py).
#ifndef LSDR_NOTCH_DETECT_H
#define LSDR_NOTCH_DETECT_H

constexpr int kDetN = 4096;        // fft.n of auto_notch (sdr.h:55)
constexpr int kDetMaxSlots = 8;

// load(i): sample i of the detect block as float2 (cf32 items as they are; cu8 items converted like cconverter<u8,128,f32,0,1,1>)
template <typename LOAD>
__device__ __forceinline__ void cfft_half_body_t(LOAD load, const float2 *om, float2 *halves /*[…][2][2048]*/, unsigned half_index) {
  __shared__ float2 d[kDetN / 2];
  const int h = half_index & 1, tid = threadIdx.x;
  for (int p = tid; p < kDetN / 2; p += 256) d[p] = load(__brev((unsigned)(h * (kDetN / 2) + p)) >> 20);   // position P holds in[brev12(P)]
  __syncthreads();
  for (int st = 0; st < 11; ++st) {
    const int hbs = 1 << st, dom = 1 << (11 - st);
    for (int b = tid; b < kDetN / 4; b += 256) {
      const int j = b >> st, k = b & (hbs - 1);
      const int p = j * hbs * 2 + k, q = p + hbs;
      const float2 w = om[k * dom], dd = d[q], dp = d[p];
      const float xr = w.x * dd.x - w.y * dd.y;
      const float xi = w.x * dd.y + w.y * dd.x;
      d[q] = make_float2(dp.x - xr, dp.y - xi);
      d[p] = make_float2(dp.x + xr, dp.y + xi);
    }
    __syncthreads();
  }
  for (int p = tid; p < kDetN / 2; p += 256) halves[(size_t)half_index * (kDetN / 2) + p] = d[p];
}

// peak search of detect() (sdr.h:94-117) on spectrum `spec_index` (one workgroup of 256): amplitudes by hypotf, nslots rounds of
// "first maximum wins, zero it and its two neighbours"
__device__ __forceinline__ void notch_peaks_body(const float2 *halves, const float2 *om, float invn, int nslots,
                                                 int *cand /*[…][kDetMaxSlots]*/, unsigned spec_index) {
  __shared__ float amp[kDetN];
  __shared__ float s_v[256];
  __shared__ int s_i[256];
  const float2 *he = halves + (size_t)spec_index * kDetN, *ho = he + kDetN / 2;
  for (int k = threadIdx.x; k < kDetN / 2; k += 256) {   // last radix-2 stage + the reverse transform's 1/n + |.|
    const float2 w = om[k], dd = ho[k], dp = he[k];
    const float xr = w.x * dd.x - w.y * dd.y;
    const float xi = w.x * dd.y + w.y * dd.x;
    amp[k + kDetN / 2] = hypotf((dp.x - xr) * invn, (dp.y - xi) * invn);
    amp[k] = hypotf((dp.x + xr) * invn, (dp.y + xi) * invn);
  }
  __syncthreads();
  for (int s = 0; s < nslots; ++s) {
    float bv = -1.f; int bi = 0;
    for (int i = threadIdx.x; i < kDetN; i += 256) if (amp[i] > bv) { bv = amp[i]; bi = i; }     // ascending i per lane: first max kept
    s_v[threadIdx.x] = bv; s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
      if ((int)threadIdx.x < d) {
        const float ov = s_v[threadIdx.x + d]; const int oi = s_i[threadIdx.x + d];
        if (ov > s_v[threadIdx.x] || (ov == s_v[threadIdx.x] && oi < s_i[threadIdx.x])) { s_v[threadIdx.x] = ov; s_i[threadIdx.x] = oi; }
      }
      __syncthreads();
    }
    const int im = s_i[0];
    if (threadIdx.x == 0) {
      cand[spec_index * kDetMaxSlots + s] = im;
      amp[im] = 0;
      if (im - 1 >= 0) amp[im - 1] = 0;
      if (im + 1 < kDetN) amp[im + 1] = 0;
    }
    __syncthreads();
  }
}

// cfft_engine's twiddles (dsp.h:62-72): omega[i] = (cosf(a), ∓sinf(a)), a = (float)(2π·i/n) — built on the host with libm like the reference
static inline void notch_detect_twiddles(int n, bool reverse, std::vector<float2> &om) {
  om.resize(n);
  for (int i = 0; i < n; ++i) {
    const float a = (float)(2.0 * M_PI * i / n);
    om[i].x = cosf(a);
    om[i].y = reverse ? -sinf(a) : sinf(a);
  }
}

#endif  // LSDR_NOTCH_DETECT_H
