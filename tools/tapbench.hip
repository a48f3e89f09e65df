// tools/tapbench.hip — isolates the tap phase of k_fir_persist to find what bounds it.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/tapbench.hip -o tools/tapbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) float *cptr1;
constexpr int D = 30, S = 269, NCOL = 11, REP = 16;
typedef unsigned v2u __attribute__((ext_vector_type(2)));
#include "tapbench_asm.inc"

// VAR 0: s_load coeffs + ds_read (as shipped)   1: coeffs from kernel-arg constants (no s_load in loop)
// VAR 2: s_load coeffs, no LDS (register sample) 3: ds_read only, coefficient = 1 literal
// VAR 4: like 0 but two independent accumulators (even/odd taps)  -> NOT reference order, only a probe
// PF: 32 buffer loads per lane in flight during the tap phase (as in k_fir_persist)
// VAR 5: coefficients staged in LDS, read with uniform-address ds_read_b128
// VAR 6: coefficients held in 6 VGPRs (lane i%64 of register i/64), fetched with v_readlane
template <int VAR, int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k(const float *rc, float2 *out, unsigned long long *cyc, float kc, const float2 *big, size_t nbig) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2 *lds = reinterpret_cast<float2 *>(smem);
  const unsigned l = threadIdx.x;
  for (unsigned t = l; t < (unsigned)(D * S); t += 256) lds[t] = make_float2(t * 0.001f, 1.0f - t * 0.002f);
  float *lc = reinterpret_cast<float *>(lds + D * S);   // coefficient copy in LDS (VAR 5)
  for (unsigned t = l; t < (unsigned)(NCOL * D + 2); t += 256) lc[t] = rc[t];
  float creg[6];
  for (int i = 0; i < 6; ++i) creg[i] = rc[i * 64 + (l & 63)];
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float2 acc = make_float2(0.f, 0.f), acc2 = make_float2(0.f, 0.f);
  float2 pf[32];
  for (int rep = 0; rep < REP; ++rep) {
    if (PF) {
      size_t j0 = ((size_t)(blockIdx.x * REP + rep) * 8192) % (nbig - 8192);
      __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(big + j0), 0, 65536, 0x00020000);
#pragma unroll
      for (int q = 0; q < 32; ++q) { v2u r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, l * 8u, q * 2048, 0); pf[q] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y)); }
      __builtin_amdgcn_sched_barrier(0);
    }
    cptr1 prc = (cptr1)rc;
    if (VAR == 8) {   // hand-scheduled 16 + 14 tap blocks, coefficients of the next block prefetched inside the block
      typedef const __attribute__((address_space(4))) v2f *cpair;
      cpair pcp = (cpair)rc;
      v2f a0 = pcp[0], a1 = pcp[1], a2 = pcp[2], a3 = pcp[3], a4 = pcp[4], a5 = pcp[5], a6 = pcp[6], a7 = pcp[7];
      v2f b0, b1, b2, b3, b4, b5, b6;
      v2f ac = {acc.x, acc.y};
      for (int col = NCOL - 1; col >= 0; --col) {
        const int ci = NCOL - 1 - col;
        const unsigned addr = (l + (unsigned)col) * 8u;
        taps_a16<D, S>(ac, addr, (const void __attribute__((address_space(4))) *)(pcp + ci * 15 + 8), a0, a1, a2, a3, a4, a5, a6, a7, b0, b1, b2, b3,
                       b4, b5, b6);
        taps_b14<D, S>(ac, addr, (const void __attribute__((address_space(4))) *)(pcp + (ci + 1) * 15), b0, b1, b2, b3, b4, b5, b6, a0, a1, a2, a3, a4,
                       a5, a6, a7);
      }
      acc.x = ac.x; acc.y = ac.y;
    } else if (VAR == 7) {   // two taps per ds_read_b128: rows interleaved in 16-byte units
      for (int col = NCOL - 1; col >= 0; --col, prc += D) {
        const float4 *px4 = reinterpret_cast<const float4 *>(lds) + l + (D / 2 - 1) * S + (unsigned)col;
        float4 v[D / 2];
#pragma unroll
        for (int k = 0; k < D / 2; ++k) v[k] = px4[-k * S];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < D / 2; ++k) {
          const float c0 = prc[2 * k], c1 = prc[2 * k + 1];
          acc.x = acc.x + c0 * v[k].z; acc.y = acc.y + c0 * v[k].w;
          acc.x = acc.x + c1 * v[k].x; acc.y = acc.y + c1 * v[k].y;
        }
      }
    } else
    for (int col = NCOL - 1; col >= 0; --col, prc += D) {
      const float2 *px = lds + l + (D - 1) * S + (unsigned)col;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const int ti = (NCOL - 1 - col) * D + k;   // tap index
        float c;
        if (VAR == 1) c = kc * (k + 1);
        else if (VAR == 3) c = 1.0f;
        else if (VAR == 5) c = lc[ti];
        else if (VAR == 6) c = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(creg[(ti >> 6) % 6]), ti & 63));
        else c = prc[k];
        float2 x = (VAR == 2) ? make_float2(acc.y, acc.x) : px[-k * S];
        if (VAR == 4 && (k & 1)) { acc2.x = acc2.x + c * x.x; acc2.y = acc2.y + c * x.y; }
        else { acc.x = acc.x + c * x.x; acc.y = acc.y + c * x.y; }
      }
    }
    if (PF) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 32; ++q) { acc2.x += pf[q].x; acc2.y += pf[q].y; }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + l] = make_float2(acc.x + acc2.x, acc.y + acc2.y);
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}
static float2 *g_big; static size_t g_nbig;
template <int VAR, int PF = 0> void run(const char *name, int grid, const float *rc, float2 *out, unsigned long long *cyc) {
  size_t lds = (size_t)D * S * 8 + 2048;
  hipLaunchKernelGGL((k<VAR, PF>), dim3(grid), dim3(256), lds, 0, rc, out, cyc, 0.001f, g_big, g_nbig);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  printf("%-44s grid %4d: %8.0f cycles/tile-equivalent (%.1f per tap)\n", name, grid, s / grid / REP, s / grid / REP / (NCOL * D));
}
int main() {
  float *rc; float2 *out; unsigned long long *cyc;
  hipMalloc(&rc, 4096); hipMalloc(&out, 2048 * 256 * 8); hipMalloc(&cyc, 2048 * 8);
  g_nbig = 64ull << 20; hipMalloc(&g_big, g_nbig * 8); hipMemset(g_big, 0, g_nbig * 8);
  std::vector<float> h(1024, 0.003f); hipMemcpy(rc, h.data(), 4096, hipMemcpyHostToDevice);
  for (int grid : {256, 512}) {
    run<0>("s_load coeffs + ds_read_b64 (shipped)", grid, rc, out, cyc);
    run<1>("arith coeffs (no s_load) + ds_read_b64", grid, rc, out, cyc);
    run<2>("s_load coeffs, no LDS", grid, rc, out, cyc);
    run<3>("ds_read_b64 only, c=1", grid, rc, out, cyc);
    run<4>("shipped + 2 accumulators (probe)", grid, rc, out, cyc);
    run<0, 1>("shipped + 32 loads in flight", grid, rc, out, cyc);
    run<3, 1>("ds_read only c=1 + 32 loads in flight", grid, rc, out, cyc);
    run<2, 1>("s_load only (no LDS) + 32 loads in flight", grid, rc, out, cyc);
    run<5>("coeffs in LDS (uniform ds_read)", grid, rc, out, cyc);
    run<5, 1>("coeffs in LDS + 32 loads in flight", grid, rc, out, cyc);
    run<6>("coeffs via v_readlane", grid, rc, out, cyc);
    run<6, 1>("coeffs via v_readlane + 32 loads in flight", grid, rc, out, cyc);
    run<7>("s_load coeffs + ds_read_b128 (2 taps)", grid, rc, out, cyc);
    run<7, 1>("ds_read_b128 (2 taps) + 32 loads in flight", grid, rc, out, cyc);
    {   // same taps, same order: the asm blocks must reproduce the compiler's result bit for bit
      std::vector<float2> ra(grid * 256), rb(grid * 256);
      run<0>("(again) shipped", grid, rc, out, cyc);
      hipMemcpy(ra.data(), out, ra.size() * 8, hipMemcpyDeviceToHost);
      run<8>("asm 16+14 blocks, coeff prefetch", grid, rc, out, cyc);
      hipMemcpy(rb.data(), out, rb.size() * 8, hipMemcpyDeviceToHost);
      printf("asm blocks vs compiler: %s\n", memcmp(ra.data(), rb.data(), ra.size() * 8) ? "DIFFERENT" : "bit-identical");
    }
    run<8, 1>("asm 16+14 blocks + 32 loads in flight", grid, rc, out, cyc);
  }
  return 0;
}
