// leansdr_amd/csrc/elementwise.hip — pure streaming maps (HBM-bound).
//   cconverter<u8,128,f32,0,1,1>   dsp.h:33-54
//   scaler<float,cf32,cf32>        dsp.h:140-160
//   decimator<cf32>                generic.h:247-267
// 16 B per lane per access, grid-stride, grid capped at 8 blocks/CU
// (cdna_hip_programming.md G11/G13).  Stand-alone forms of the stages that
// fir_filter / cstln_receiver can also fuse into their loads.
#include "lsdr_internal.h"

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(const lsdr_ctx *c, size_t work_items) {
  size_t blocks = (work_items + kBlock - 1) / kBlock;
  size_t cap = (size_t)c->num_cu * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// 4 samples (8 bytes of cu8) per lane-iteration -> 2 x float4 stores.
__global__ __launch_bounds__(kBlock) void k_cconv_u8(const uint8_t *__restrict__ in, size_t n,
                                                      float *__restrict__ out) {
  size_t nvec = n / 4;  // groups of 4 complex samples
  size_t stride = (size_t)gridDim.x * kBlock;
  const bool aligned = (((uintptr_t)in) & 7) == 0 && (((uintptr_t)out) & 15) == 0;
  if (aligned) {
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
      uint2 raw = *reinterpret_cast<const uint2 *>(in + v * 8);
      float4 a, b;
      // out = 0 + (in - 128)*1/1 evaluated in int, then converted (dsp.h:46-47)
      a.x = (float)((int)(raw.x & 0xff) - 128);
      a.y = (float)((int)((raw.x >> 8) & 0xff) - 128);
      a.z = (float)((int)((raw.x >> 16) & 0xff) - 128);
      a.w = (float)((int)(raw.x >> 24) - 128);
      b.x = (float)((int)(raw.y & 0xff) - 128);
      b.y = (float)((int)((raw.y >> 8) & 0xff) - 128);
      b.z = (float)((int)((raw.y >> 16) & 0xff) - 128);
      b.w = (float)((int)(raw.y >> 24) - 128);
      float4 *o = reinterpret_cast<float4 *>(out + v * 8);
      o[0] = a;
      o[1] = b;
    }
  } else {
    nvec = 0;
  }
  // tail (and the whole range when misaligned)
  for (size_t i = nvec * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    out[2 * i] = (float)((int)in[2 * i] - 128);
    out[2 * i + 1] = (float)((int)in[2 * i + 1] - 128);
  }
}

// out = in * scale, complex*T = (re*k, im*k)  (math.h:45-48): one multiply per float.
__global__ __launch_bounds__(kBlock) void k_scale(const float *__restrict__ in, size_t nfloat, float scale,
                                                   float *__restrict__ out) {
  size_t stride = (size_t)gridDim.x * kBlock;
  size_t nvec = 0;
  if ((((uintptr_t)in) & 15) == 0 && (((uintptr_t)out) & 15) == 0) {
    nvec = nfloat / 4;
    for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
      float4 x = reinterpret_cast<const float4 *>(in)[v];
      x.x *= scale; x.y *= scale; x.z *= scale; x.w *= scale;
      reinterpret_cast<float4 *>(out)[v] = x;
    }
  }
  for (size_t i = nvec * 4 + (size_t)blockIdx.x * kBlock + threadIdx.x; i < nfloat; i += stride)
    out[i] = in[i] * scale;
}

__global__ __launch_bounds__(kBlock) void k_decim(const float2 *__restrict__ in, unsigned d, size_t count,
                                                   float2 *__restrict__ out) {
  size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t m = (size_t)blockIdx.x * kBlock + threadIdx.x; m < count; m += stride) out[m] = in[m * d];
}

// rotator<f32>, sdr.h:1243-1254: out = in·(cos, sin)[index], 16-bit table index advancing by one per sample.
__global__ __launch_bounds__(kBlock) void k_rotate(const float2 *__restrict__ in, size_t n, const float2 *__restrict__ lut,
                                                   unsigned index0, float2 *__restrict__ out) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const float2 cs = lut[(index0 + (unsigned)(i & 0xffffu)) & 0xffffu];
    const float2 x = in[i];
    out[i] = make_float2(x.x * cs.x - x.y * cs.y, x.x * cs.y + x.y * cs.x);
  }
}

}  // namespace

// cconverter<s8,0,f32,0,1,1>, <u16,32768,f32,0,1,1>, <s16,0,f32,0,1,1> (leandvb --s8/--u16/--s16, leandvb.cc:218-248):
// out = 0 + (in − Zin)·1/1 in int arithmetic, then int → float (dsp.h:46-47).  One scalar component per lane-step,
// 4 components (two complex samples) per lane: coalesced 4/8-byte loads, 16-byte stores.
template <typename Tin, int Zin>
__global__ __launch_bounds__(kBlock) void k_cconv_int(const Tin *__restrict__ in, size_t ncomp, float *__restrict__ out) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  const size_t nvec = ncomp / 4;
  for (size_t v = (size_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
    Tin r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = in[v * 4 + k];
    float4 o;
    o.x = (float)((int)r[0] - (int)(Tin)Zin); o.y = (float)((int)r[1] - (int)(Tin)Zin);
    o.z = (float)((int)r[2] - (int)(Tin)Zin); o.w = (float)((int)r[3] - (int)(Tin)Zin);
    if ((((uintptr_t)out) & 15) == 0) *reinterpret_cast<float4 *>(out + v * 4) = o;
    else { out[v * 4] = o.x; out[v * 4 + 1] = o.y; out[v * 4 + 2] = o.z; out[v * 4 + 3] = o.w; }
  }
  if (blockIdx.x == 0 && threadIdx.x < ncomp - nvec * 4)
    out[nvec * 4 + threadIdx.x] = (float)((int)in[nvec * 4 + threadIdx.x] - (int)(Tin)Zin);
}

extern "C" {

int lsdr_cconverter_u8_run(lsdr_ctx *c, const lsdr_cu8 *in, size_t n, lsdr_cf32 *out) {
  LSDR_ARG(c && (n == 0 || (in && out)));
  if (!n) return LSDR_OK;
  hipLaunchKernelGGL(k_cconv_u8, dim3(grid_for(c, n / 4 + 1)), dim3(kBlock), 0, c->stream,
                     (const uint8_t *)in, n, (float *)out);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

int lsdr_cconverter_int_run(lsdr_ctx *c, int in_format, const void *in, size_t n, lsdr_cf32 *out) {
  LSDR_ARG(c && (n == 0 || (in && out)));
  if (!n) return LSDR_OK;
  const dim3 g(grid_for(c, n / 2 + 1)), b(kBlock);
  switch (in_format) {
    case LSDR_IN_CS8: hipLaunchKernelGGL((k_cconv_int<int8_t, 0>), g, b, 0, c->stream, (const int8_t *)in, 2 * n, (float *)out); break;
    case LSDR_IN_CU16: hipLaunchKernelGGL((k_cconv_int<uint16_t, 32768>), g, b, 0, c->stream, (const uint16_t *)in, 2 * n, (float *)out); break;
    case LSDR_IN_CS16: hipLaunchKernelGGL((k_cconv_int<int16_t, 0>), g, b, 0, c->stream, (const int16_t *)in, 2 * n, (float *)out); break;
    default: lsdr_set_error("cconverter: unsupported input format %d", in_format); return LSDR_E_ARG;
  }
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

int lsdr_scaler_run(lsdr_ctx *c, float scale, const lsdr_cf32 *in, size_t n, lsdr_cf32 *out) {
  LSDR_ARG(c && (n == 0 || (in && out)));
  if (!n) return LSDR_OK;
  hipLaunchKernelGGL(k_scale, dim3(grid_for(c, n / 2 + 1)), dim3(kBlock), 0, c->stream, (const float *)in,
                     2 * n, scale, (float *)out);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

int lsdr_decimator_run(lsdr_ctx *c, unsigned d, const lsdr_cf32 *in, size_t n, lsdr_cf32 *out, size_t cap,
                       size_t *produced) {
  LSDR_ARG(c && d >= 1 && produced);
  size_t count = n / d;
  if (count > cap) count = cap;
  *produced = count;
  if (!count) return LSDR_OK;
  LSDR_ARG(in && out);
  hipLaunchKernelGGL(k_decim, dim3(grid_for(c, count)), dim3(kBlock), 0, c->stream, (const float2 *)in, d,
                     count, (float2 *)out);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

// ------------------------------------------------------------------ rotator<f32> (sdr.h:1226-1259)
struct lsdr_rotator { lsdr_ctx *ctx; float2 *d_lut; unsigned index; };

int lsdr_rotator_create(lsdr_ctx *c, float freq, lsdr_rotator **out) {
  LSDR_ARG(c && out);
  LSDR_HIP(hipSetDevice(c->device));
  std::vector<float2> lut(65536);
  const int ifreq = (int)(freq * 65536);                    // sdr.h:1231
  for (int i = 0; i < 65536; ++i) {                         // host libm, like the reference (sdr.h:1235-1238)
    lut[i].x = cosf(2 * M_PI * i * ifreq / 65536);
    lut[i].y = sinf(2 * M_PI * i * ifreq / 65536);
  }
  lsdr_rotator *r = new lsdr_rotator();
  r->ctx = c; r->index = 0;
  LSDR_HIP(hipMalloc((void **)&r->d_lut, lut.size() * sizeof(float2)));
  LSDR_HIP(hipMemcpy(r->d_lut, lut.data(), lut.size() * sizeof(float2), hipMemcpyHostToDevice));
  *out = r;
  return LSDR_OK;
}
void lsdr_rotator_destroy(lsdr_rotator *r) {
  if (!r) return;
  (void)hipStreamSynchronize(r->ctx->stream);
  (void)hipFree(r->d_lut);
  delete r;
}
int lsdr_rotator_run(lsdr_rotator *r, const lsdr_cf32 *in, size_t n, lsdr_cf32 *out) {
  LSDR_ARG(r && (n == 0 || (in && out)));
  if (!n) return LSDR_OK;
  hipLaunchKernelGGL(k_rotate, dim3(grid_for(r->ctx, n)), dim3(kBlock), 0, r->ctx->stream, (const float2 *)in, n,
                     (const float2 *)r->d_lut, r->index, (float2 *)out);
  LSDR_HIP(hipGetLastError());
  r->index = (r->index + (unsigned)(n & 0xffffu)) & 0xffffu;
  return LSDR_OK;
}

}  // extern "C"
