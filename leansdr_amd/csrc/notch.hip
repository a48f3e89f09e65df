// leansdr_amd/csrc/notch.hip — auto_notch<f32> (sdr.h:46-154), cnr_fft<f32> (sdr.h:1273-1345) and the
// host restatement of cfft_engine<float> (dsp.h:56-116) they share.
//
// auto_notch::process() is, per slot, the recurrence   estim ← bb·k + estim·(1−k)   over every sample
// (bb = x·conj(e_i), e_i restarts every 4096-sample block) followed by  out = gain·(x − Σ estim·e_i).
// Float rounding makes a scan formulation inexact, so each GPU lane runs the reference's sequential
// arithmetic over a time tile.  Lanes other than the first start `kWarmBlocks` blocks early from
// estim = 0: the influence of the start value decays by (1−k) per sample (0.998^16384 ≈ 6e-15) and the
// two trajectories then coincide BIT FOR BIT — which is verified at every seam (tile j's estimators
// after warm-up == tile j−1's estimators at its end); anything that fails is redone sequentially.
// Memory-bound streaming: 8 B in + 8 B out per sample; neighbouring lanes read different cache lines
// but each line is reused by its lane for 16 consecutive samples (L1-resident).
#include <cmath>
#include "lsdr_internal.h"

namespace {
#include "notch_detect.h"

constexpr int kN = 4096;          // fft.n of auto_notch (sdr.h:55)
constexpr int kMaxSlots = 8;
constexpr int kTrip = 32;          // samples per register trip of k_notch
constexpr int kTileBlocks = 1, kWarmBlocks = 4;   // forgetting the start takes ≈ 8300 samples (0.998^n below float resolution) plus a few thousand for two nearby roundings to coincide; 3 blocks fail verification about once per run

struct notch_est { float re[kMaxSlots], im[kMaxSlots]; };

struct notch_args {
  const float2 *in;
  float2 *out;
  const float2 *expj;              // [nslots][4096]
  int nslots;
  float k, gain;
  unsigned long long n_blocks;     // blocks of this launch
  unsigned tile_blocks, warm_blocks;
  unsigned n_tiles;
  const notch_est *carry;          // estimators at block 0
  notch_est *begin, *end;          // per tile
  int serial_from;                 // ≥ 0: single sequential job starting at this block (repair); -1: tiled
};

typedef float notch_v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) notch_v2f *notch_cptr;

constexpr int kLdsSlots = 4;   // 32 KB of phasors per slot
// Tiles (lanes) per wavefront.  Every lane walks its own stretch of memory, so a load/store instruction touches one cache
// line per active lane and the address unit serialises them: with 64 lanes the 32 memory instructions of a trip cost
// ≈ 64 cycles each (≈ 64 cycles/sample, twice the arithmetic).  The GPU is otherwise idle here (a run has a few hundred
// tiles), so fewer lanes per wave and more waves is free.
constexpr int kNotchLanes = 16;

template <int NS>
__global__ __launch_bounds__(64) void k_notch(notch_args a) {   // 64 threads fill the LDS, kNotchLanes of them own a tile
  extern __shared__ __attribute__((aligned(16))) char notch_smem[];
  notch_v2f *lds_e = reinterpret_cast<notch_v2f *>(notch_smem);
  if (NS <= kLdsSlots) {
    for (int i = threadIdx.x; i < NS * kN; i += 64) { const float2 v = a.expj[i]; notch_v2f w = {v.x, v.y}; lds_e[i] = w; }
    __syncthreads();
  }
  if (threadIdx.x >= kNotchLanes) return;
  const unsigned t = blockIdx.x * kNotchLanes + threadIdx.x;
  if (t >= a.n_tiles) return;
  unsigned long long b0, b1;
  bool from_carry;
  if (a.serial_from >= 0) { b0 = (unsigned long long)a.serial_from; b1 = a.n_blocks; from_carry = true; }
  else if (t == 0) {   // tile 0 is long enough (warm-up + tile) for tile 1 to warm up fully inside this launch
    b0 = 0; b1 = (unsigned long long)a.warm_blocks + a.tile_blocks;
    if (b1 > a.n_blocks) b1 = a.n_blocks;
    from_carry = true;
  } else {
    b0 = (unsigned long long)a.warm_blocks + a.tile_blocks + (unsigned long long)(t - 1) * a.tile_blocks;
    b1 = b0 + a.tile_blocks;
    if (b1 > a.n_blocks) b1 = a.n_blocks;
    from_carry = false;
  }
  float er[NS], ei[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { er[s] = from_carry ? a.carry->re[s] : 0.f; ei[s] = from_carry ? a.carry->im[s] : 0.f; }
  const float k = a.k, omk = 1 - a.k, gain = a.gain;
  unsigned long long bstart = b0;
  if (!from_carry) bstart = b0 - a.warm_blocks;
  for (unsigned long long b = bstart; b < b1; ++b) {
    if (b == b0) {
#pragma unroll
      for (int s = 0; s < NS; ++s) { a.begin[t].re[s] = er[s]; a.begin[t].im[s] = ei[s]; }
    }
    const bool emit = b >= b0;
    const float2 *pin = a.in + b * kN;
    float2 *pout = a.out + b * kN;
    // 8 samples per trip: the loads (one 64-byte line per lane) and the phasor fetches (wave-uniform) are issued
    // together, the recurrence then runs on registers, the 8 results leave as one line.  The per-sample
    // arithmetic and its order are the reference's (sdr.h:124-134).
    // Two trips (kTrip samples each) are kept in flight in alternating register sets.  gfx9 counts loads and stores in
    // the same vmcnt and cannot wait on loads past younger stores, so every wait for a trip's samples also drains the
    // previous trip's stores: long trips amortise that round trip (8-sample trips: ~120 cycles/sample, mostly drain).
    auto trip = [&](const float2 (&x8)[kTrip], int i0) {
      float2 o8[kTrip];
#pragma unroll
      for (int j = 0; j < kTrip; ++j) {
        // (re, im) pairs as 2-vectors → v_pk_mul_f32 / v_pk_add_f32: the same individually rounded products and sums
        // as the scalar expressions of sdr.h:124-134 (negations are exact), half the instructions.
        const notch_v2f xx = {x8[j].x, x8[j].x}, xy = {x8[j].y, x8[j].y};
        notch_v2f o = {x8[j].x, x8[j].y};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          // slot phasors: wave-uniform.  LDS copy (broadcast ds_read, pipelined by the scheduler) when it fits, else s_load
          const notch_v2f e = NS <= kLdsSlots ? lds_e[s * kN + i0 + j] : ((notch_cptr)a.expj)[s * kN + i0 + j];
          const notch_v2f e_conj = {e.x, -e.y}, e_swap = {e.y, e.x}, e_rot = {-e.y, e.x};
          const notch_v2f bb = xx * e_conj + xy * e_swap;              // x·conj(e)
          notch_v2f est = {er[s], ei[s]};
          est = bb * k + est * omk;
          er[s] = est.x; ei[s] = est.y;
          const notch_v2f ser = {est.x, est.x}, sei = {est.y, est.y};
          const notch_v2f sub = ser * e + sei * e_rot;                 // estim·e
          o = o - sub;
        }
        o8[j] = make_float2(gain * o.x, gain * o.y);
      }
      if (emit) {
#pragma unroll
        for (int j = 0; j < kTrip; ++j) pout[i0 + j] = o8[j];
      }
    };
    float2 xa[kTrip], xb[kTrip];
#pragma unroll
    for (int j = 0; j < kTrip; ++j) xa[j] = pin[j];
#pragma unroll
    for (int j = 0; j < kTrip; ++j) xb[j] = pin[kTrip + j];
    for (int i0 = 0; i0 < kN; i0 += 2 * kTrip) {
      trip(xa, i0);
      if (i0 + 2 * kTrip < kN) {
#pragma unroll
        for (int j = 0; j < kTrip; ++j) xa[j] = pin[i0 + 2 * kTrip + j];
      }
      trip(xb, i0 + kTrip);
      if (i0 + 3 * kTrip < kN) {
#pragma unroll
        for (int j = 0; j < kTrip; ++j) xb[j] = pin[i0 + 3 * kTrip + j];
      }
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) { a.end[t].re[s] = er[s]; a.end[t].im[s] = ei[s]; }
}

__global__ __launch_bounds__(256) void k_notch_passthrough(const float2 *in, float2 *out, size_t n, float gain) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float2 x = in[i];
    out[i] = make_float2(gain * x.x, gain * x.y);   // no active slot: out = gain·in (x − 0 = x)
  }
}

// cfft_engine<float>::inplace, dsp.h:56-116 — host, exact.
void cfft_host(int n, lsdr_cf32 *data, bool reverse) {
  int logn = 0;
  for (int t = n; t > 1; t >>= 1) ++logn;
  std::vector<lsdr_cf32> om(n);
  for (int i = 0; i < n; ++i) {
    float a = (float)(2.0 * M_PI * i / n);
    om[i].re = cosf(a);
    om[i].im = reverse ? -sinf(a) : sinf(a);
  }
  for (int i = 0; i < n; ++i) {
    int r = 0;
    for (int b = 0; b < logn; ++b) r = (r << 1) | ((i >> b) & 1);
    if (r < i) { lsdr_cf32 tmp = data[i]; data[i] = data[r]; data[r] = tmp; }
  }
  for (int i = 0; i < logn; ++i) {
    const int hbs = 1 << i, dom = 1 << (logn - 1 - i);
    for (int j = 0; j < dom; ++j) {
      const int p = j * hbs * 2, q = p + hbs;
      for (int k = 0; k < hbs; ++k) {
        const lsdr_cf32 w = om[k * dom], d = data[q + k];
        const float xr = w.re * d.re - w.im * d.im;
        const float xi = w.re * d.im + w.im * d.re;
        data[q + k].re = data[p + k].re - xr;
        data[q + k].im = data[p + k].im - xi;
        data[p + k].re = data[p + k].re + xr;
        data[p + k].im = data[p + k].im + xi;
      }
    }
  }
  if (reverse) {
    const float invn = (float)(1.0 / n);
    for (int i = 0; i < n; ++i) { data[i].re *= invn; data[i].im *= invn; }
  }
}

// cfft_engine<float>::inplace on the GPU: one workgroup, the whole transform in LDS, one barrier per radix-2 stage.
// Every butterfly is the reference's expression (dsp.h:96-104) evaluated once, so the result is bit-identical to
// cfft_host(); the twiddles om[] are the host-built cosf/sinf table of the engine's constructor (dsp.h:62-70).
__global__ __launch_bounds__(1024) void k_cfft(const float2 *in, const float2 *om, float2 *out, int logn, int reverse, float invn,
                                                const unsigned long long *in_offsets = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char cfft_smem[];
  float2 *d = reinterpret_cast<float2 *>(cfft_smem);
  const int n = 1 << logn, tid = threadIdx.x, nt = blockDim.x;   // any workgroup size up to 1024
  float2 *w_lds = d + n;                                           // the n/2 twiddles the butterflies use (om[0 .. n/2))
  if (in_offsets) { in += in_offsets[blockIdx.x]; out += (size_t)blockIdx.x * n; }   // batched: one transform per workgroup
  for (int i = tid; i < n; i += nt) d[__brev((unsigned)i) >> (32 - logn)] = in[i];   // bit-reversal permutation (dsp.h:84-92)
  // Twiddles into LDS up front: a global load per butterfly inside the stage loop pays the memory latency logn times, and next
  // to kernels that saturate HBM that latency is microseconds (the batched detect FFTs took 88-91 us there, 15 alone).
  for (int i = tid; i < n / 2; i += nt) w_lds[i] = om[i];
  __syncthreads();
  for (int st = 0; st < logn; ++st) {
    const int hbs = 1 << st, dom = 1 << (logn - 1 - st);
    for (int b = tid; b < n / 2; b += nt) {
      const int j = b >> st, k = b & (hbs - 1);
      const int p = j * hbs * 2 + k, q = p + hbs;
      const float2 w = w_lds[k * dom], dd = d[q], dp = d[p];
      const float xr = w.x * dd.x - w.y * dd.y;
      const float xi = w.x * dd.y + w.y * dd.x;
      d[q] = make_float2(dp.x - xr, dp.y - xi);
      d[p] = make_float2(dp.x + xr, dp.y + xi);
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += nt) {
    float2 v = d[i];
    if (reverse) { v.x *= invn; v.y *= invn; }
    out[i] = v;
  }
}

// Per-(n, direction) device resources of the FFT: twiddle table + output scratch.
struct cfft_dev {
  int n, logn, reverse;
  float2 *d_om, *d_out;
  cfft_dev() : n(0), logn(0), reverse(0), d_om(nullptr), d_out(nullptr) {}
};
static int cfft_dev_init(cfft_dev *f, int n, bool reverse) {
  if (f->d_om && f->n == n && f->reverse == (int)reverse) return LSDR_OK;
  (void)hipFree(f->d_om); (void)hipFree(f->d_out);
  f->n = n; f->reverse = reverse; f->logn = 0;
  for (int t = n; t > 1; t >>= 1) ++f->logn;
  std::vector<float2> om;
  notch_detect_twiddles(n, reverse, om);
  LSDR_HIP(hipMalloc((void **)&f->d_om, (size_t)n * sizeof(float2)));
  LSDR_HIP(hipMalloc((void **)&f->d_out, (size_t)n * sizeof(float2)));
  LSDR_HIP(hipMemcpy(f->d_om, om.data(), (size_t)n * sizeof(float2), hipMemcpyHostToDevice));
  return LSDR_OK;
}
static void cfft_dev_free(cfft_dev *f) { (void)hipFree(f->d_om); (void)hipFree(f->d_out); f->d_om = f->d_out = nullptr; }
// FFT of the device block `d_in` → host `spectrum` (n complex values).
static int cfft_dev_run(lsdr_ctx *c, cfft_dev *f, const lsdr_cf32 *d_in, lsdr_cf32 *spectrum) {
  const size_t lds = (size_t)(f->n + f->n / 2) * sizeof(float2);   // data + the n/2 twiddles
  if (lds > 64 * 1024) LSDR_HIP(hipFuncSetAttribute((const void *)k_cfft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_cfft, dim3(1), dim3(1024), lds, c->stream, (const float2 *)d_in, (const float2 *)f->d_om, f->d_out, f->logn,
                     f->reverse, (float)(1.0 / f->n));
  LSDR_HIP(hipGetLastError());
  LSDR_HIP(hipMemcpyAsync(spectrum, f->d_out, (size_t)f->n * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
  LSDR_HIP(hipStreamSynchronize(c->stream));
  return LSDR_OK;
}


// ---------------------------------------------------------------------------------------- throughput mode (LSDR_NOTCH_SCAN)
// The per-slot recurrence  estim ← k·bb + (1−k)·estim  is a first-order linear recurrence with a CONSTANT pole a = 1−k,
// i.e. a scan:  estim_i = L_i + a^(i+1)·carry,  L = the same recurrence started from zero.  One wavefront per 1024-sample
// wave-block, 16 consecutive samples per lane: the recurrence in registers with the reference's two roundings per step, a
// wave scan with the powers of a, and the wave-block's carry-in by decoupled look-back over the TWELVE preceding
// wave-blocks' zero-carry totals (a^4096 = 2.7e-4, a^12288 = 2e-11: older ones are below float resolution — the mode is
// refused for a k that does not make it so), published through run-stamped flags
// so that nothing has to be cleared between runs.  With several slots the passes repeat per slot on the residual of the
// previous one, like sdr.h:124-134.  detect() (every `decimation` samples, sdr.h:66-70,76-118) stays on the device too:
// all detect points of a run depend on the INPUT only, so their FFTs run batched up front (k_cfft), k_notch_peaks does the
// hypotf peak search, k_notch_plan walks the detect points in order (bin changes reset the slot's estimator and select a new
// phasor table), k_notch_tables builds the per-interval phasor tables with the reference's expression
// (float)(2π·bin·i/4096) → cosf/sinf on the device (≤ 1 ulp from libm's).  Single pass: 8 B in + 8 B out per sample.
// NOT bit-exact (the carry terms are re-associated; device cosf/sinf/hypotf): tolerance-tested against k_notch / the oracle.
constexpr int kScanPer = 16;                 // samples per lane

struct notch_scan_consts {
  float k, omk, gain;
  float apow[kScanPer + 1];        // a^j, j = 0 … 16
  float apow16[7];                 // a^(16·2^m), m = 0 … 6   (wave scan; [6] = a^1024)
  float a1024, a2048, a3072, a4096, a8192, a12288;
};

struct notch_scan_args {
  const float2 *in;
  float2 *out;
  const float2 *tables;            // [n_intervals][nslots][4096]
  const int *interval_first;       // [n_intervals] first block of each interval (interval 0 starts at block 0)
  const unsigned char *reset;      // [n_intervals][kMaxSlots]: the slot's estimator restarts from 0 at the interval's first block
  int n_intervals, nslots;
  unsigned long long n_blocks;
  const notch_est *carry;          // estimators before block 0
  notch_est *carry_out;            // (last block) estimators after the last block → the next run's `carry`
  float2 *totals;                  // [kMaxSlots][4·n_blocks] zero-carry wave-block totals
  unsigned *flags;                 // [kMaxSlots][4·n_blocks] run stamps
  unsigned stamp;
  unsigned *abort_flag;            // pinned host word: set when a look-back wait gave up (see k_notch_scan)
  notch_scan_consts C;
};

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// Cross-workgroup hand-off of a block total: relaxed AGENT-scope atomics only (write-through store / L2-bypassing load on
// the multi-XCD part) with an explicit vmcnt(0) between the value and its stamp.  A release FENCE here would write back the
// whole L2 of the XCD — which this kernel keeps full of dirty output lines — once per block (measured: 16 K blocks in 2.9 ms
// with fences, i.e. serialised on the fence).
__device__ __forceinline__ void scan_publish(float2 *tot_slot, unsigned *flag_slot, float2 tot, unsigned stamp) {
  unsigned long long bits;
  __builtin_memcpy(&bits, &tot, 8);
  // Hand-off form "8-byte agent-scope atomics on both sides" (MI355X_MICROARCH.md, inter-workgroup visibility): the value and
  // the stamp are written through to L2 (sc1), the reader's loads bypass its L1 (sc1); the value has LEFT this CU before the
  // stamp is issued — the wait is inline asm so that no compiler pass can drop or move it.  gfx942/gfx950 semantics, not the
  // portable HIP memory model (a release fence per wave-block would flush the XCD's dirty L2: measured 12× slower).
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(tot_slot), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(flag_slot, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 scan_wait(const float2 *tot_slot, const unsigned *flag_slot, unsigned stamp, unsigned *abort_flag) {
  unsigned spins = 0;
  while (__hip_atomic_load(flag_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != stamp) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 21)) {      // ≈ 0.1 s: the predecessor is not coming (dispatch order?) — give up loudly
      unsigned expect = 0u;          // the FIRST run that gave up stays on record (later ones only inherit its garbage)
      (void)__hip_atomic_compare_exchange_strong(abort_flag, &expect, stamp, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(tot_slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float2 v;
  __builtin_memcpy(&v, &bits, 8);
  return v;
}

// One wavefront per 1024-sample wave-block (a 64-thread workgroup: nothing couples two wave-blocks but the look-back).
//  - global traffic is coalesced (each instruction moves 1 KiB of consecutive samples) and turned into the lane-blocked
//    register layout (16 consecutive samples per lane) through a padded LDS tile, both ways;
//  - pass 1 runs the recurrence with zero carry to get the lane totals, a weighted wave scan gives the wave-block total,
//    which is published; lanes 0..11 then each fetch one of the TWELVE preceding wave-block totals (a^12288 is below float
//    resolution for the reference's k) and a butterfly sum gives the carry-in;
//  - pass 2 re-runs the recurrence from the lane's carry-in — the reference's own sequential formula per lane — and
//    subtracts est·e.  (Keeping pass 1's partial sums instead costs 32 VGPRs and an occupancy step.)
constexpr int kWaveSamples = 64 * kScanPer;            // 1024
constexpr int kWavesPerBlock = kN / kWaveSamples;      // 4 wave-blocks per 4096-sample FFT block
constexpr int kLaneStride = kScanPer * 8 + 16;         // bytes between two lanes' chunks in the LDS tile (bank spread)
constexpr int kLookBack = 12;

__device__ __forceinline__ unsigned lds_off(unsigned n) { return n * 8u + (n >> 4) * 16u; }   // sample n of the wave-block

__device__ __forceinline__ void wave_load_blocked(const float2 *__restrict__ g, float2 (&r)[kScanPer], char *lds, unsigned lane) {
#pragma unroll
  for (int i = 0; i < kScanPer / 2; ++i) {
    const unsigned n = (unsigned)i * 128u + lane * 2u;
    *reinterpret_cast<float4 *>(lds + lds_off(n)) = *reinterpret_cast<const float4 *>(g + n);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kScanPer / 2; ++k) {
    const float4 v = *reinterpret_cast<const float4 *>(lds + lane * kLaneStride + k * 16);
    r[2 * k] = make_float2(v.x, v.y); r[2 * k + 1] = make_float2(v.z, v.w);
  }
  __syncthreads();
}

template <int NS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NS == 1 ? 4 : 2))) void k_notch_scan(notch_scan_args a) {
  __shared__ __attribute__((aligned(16))) char lds[64 * kLaneStride];
  const unsigned lane = threadIdx.x;
  // Wave-block = blockIdx: a wave-block spins on its twelve predecessors' totals, and those belong to LOWER block indices, which
  // gfx942 / gfx950 start first (observed; promised nowhere).  Two dispatch-order-independent forms were built and measured on
  // the anf1 pipeline (k_notch_scan 0.22 ms per 64 Mi samples as written here): a ticket per workgroup — 65 536 agent-scope
  // atomics on one address per launch — 0.83 ms; persistent workgroups that take one ticket each and walk wave-blocks
  // t, t + grid, … 0.58 – 1.1 ms (4 – 16 workgroups per CU).  So the order stays an ASSUMPTION, and it is made safe instead of
  // fast-and-hopeful: every spin is bounded (scan_wait gives up after ≈ 0.1 s), a wave-block that gave up raises the run's abort
  // word in pinned memory, and the host refuses to go on (lsdr_auto_notch_run returns an error at the next call or at
  // lsdr_auto_notch_check) — no hang, no silently wrong output.  tests/test_gpu_notch.py poisons the hand-off buffers before each
  // of 1000 runs and demands bit-identical output.
  const unsigned long long wb = blockIdx.x, n_wb = a.n_blocks * kWavesPerBlock;
  {
  const unsigned long long b = wb / kWavesPerBlock;
  const unsigned part = (unsigned)(wb % kWavesPerBlock);
  // interval of this block (few intervals: linear search, wave-uniform)
  int q = 0;
  while (q + 1 < a.n_intervals && (unsigned long long)a.interval_first[q + 1] <= b) ++q;
  const notch_scan_consts &C = a.C;
  float2 x[kScanPer], o[NS > 1 ? kScanPer : 1];
  wave_load_blocked(a.in + wb * kWaveSamples, x, lds, lane);
  if (NS > 1) {
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) o[NS > 1 ? j : 0] = x[j];
  }
  float pl = 1.f;                      // a^(16·lane) from the binary powers
#pragma unroll
  for (int m = 0; m < 6; ++m) if (lane & (1u << m)) pl *= C.apow16[m];
  float wl = 1.f;                      // a^(1024·lane), lanes 0..11: the weight of the (lane+1)-th wave-block back
  if (lane & 1u) wl *= C.a1024;
  if (lane & 2u) wl *= C.a2048;
  if (lane & 4u) wl *= C.a4096;
  if (lane & 8u) wl *= C.a8192;
  // every slot filters the RAW input (sdr.h:126-128: bb from *pin); the slots' corrections are subtracted in slot order
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float2 e[kScanPer];
    wave_load_blocked(a.tables + ((size_t)q * a.nslots + s) * kN + part * kWaveSamples, e, lds, lane);
    float2 run = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
      const float2 bb = make_float2(x[j].x * e[j].x + x[j].y * e[j].y, -x[j].x * e[j].y + x[j].y * e[j].x);   // x·conj(e)
      run = make_float2(bb.x * C.k + run.x * C.omk, bb.y * C.k + run.y * C.omk);
    }
    // inclusive scan of the lane totals across the wave: P_lane = Σ_{u ≤ lane} a^(16(lane−u))·T_u
    float2 P = run;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      const int d = 1 << m;
      const float ox = __shfl_up(P.x, d, 64), oy = __shfl_up(P.y, d, 64);
      if (lane >= (unsigned)d) { P.x += C.apow16[m] * ox; P.y += C.apow16[m] * oy; }
    }
    const float2 tot = make_float2(__shfl(P.x, 63, 64), __shfl(P.y, 63, 64));   // zero-carry total of the wave-block
    float2 *tslot = a.totals + (size_t)s * n_wb;
    unsigned *fslot = a.flags + (size_t)s * n_wb;
    if (lane == 0) scan_publish(tslot + wb, fslot + wb, tot, a.stamp);
    // Nothing older than the slot's latest restart (the first wave-block of an interval whose plan resets the slot) reaches
    // this wave-block; before wave-block 0 stand the carried estimators.
    long long restart = -2;
    for (int qq = 0; qq <= q; ++qq)
      if (a.reset[qq * kMaxSlots + s]) restart = (long long)a.interval_first[qq] * kWavesPerBlock;
    float2 v = make_float2(0.f, 0.f);
    if (lane < (unsigned)kLookBack) {
      const long long pb = (long long)wb - 1 - (long long)lane;
      if (pb >= 0 && pb >= restart) {
        const float2 pt = scan_wait(tslot + pb, fslot + pb, a.stamp, a.abort_flag);
        v = make_float2(wl * pt.x, wl * pt.y);
      } else if (pb == -1 && restart < 0) {
        v = make_float2(wl * a.carry->re[s], wl * a.carry->im[s]);
      }
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) { v.x += __shfl_xor(v.x, d, 64); v.y += __shfl_xor(v.y, d, 64); }
    const float2 ein = make_float2(__shfl(v.x, 0, 64), __shfl(v.y, 0, 64));
    if (wb == n_wb - 1 && lane == 0) {   // estimators after the last wave-block → next run (a separate buffer: the first ones read `carry`)
      a.carry_out->re[s] = tot.x + C.a1024 * ein.x; a.carry_out->im[s] = tot.y + C.a1024 * ein.y;
    }
    // carry into this lane: the lanes before it, then the wave-block's carry-in
    const float ox = __shfl_up(P.x, 1, 64), oy = __shfl_up(P.y, 1, 64);
    run = lane ? make_float2(ox + pl * ein.x, oy + pl * ein.y) : ein;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
      const float2 bb = make_float2(x[j].x * e[j].x + x[j].y * e[j].y, -x[j].x * e[j].y + x[j].y * e[j].x);
      run = make_float2(bb.x * C.k + run.x * C.omk, bb.y * C.k + run.y * C.omk);
      const float2 sub = cmulf(run, e[j]);
      if (NS > 1) o[NS > 1 ? j : 0] = make_float2(o[NS > 1 ? j : 0].x - sub.x, o[NS > 1 ? j : 0].y - sub.y);
      else x[j] = make_float2(x[j].x - sub.x, x[j].y - sub.y);
    }
  }
  // lane-blocked registers → LDS → coalesced stores
#pragma unroll
  for (int k = 0; k < kScanPer / 2; ++k) {
    const float2 p0 = NS > 1 ? o[NS > 1 ? 2 * k : 0] : x[2 * k], p1 = NS > 1 ? o[NS > 1 ? 2 * k + 1 : 0] : x[2 * k + 1];
    *reinterpret_cast<float4 *>(lds + lane * kLaneStride + k * 16) = make_float4(C.gain * p0.x, C.gain * p0.y, C.gain * p1.x, C.gain * p1.y);
  }
  __syncthreads();
  float2 *pout = a.out + wb * kWaveSamples;
#pragma unroll
  for (int i = 0; i < kScanPer / 2; ++i) {
    const unsigned n = (unsigned)i * 128u + lane * 2u;
    *reinterpret_cast<float4 *>(pout + n) = *reinterpret_cast<const float4 *>(lds + lds_off(n));
  }
  }
}

// Per-run argument arrays (interval starts, detect offsets) from the kernel-argument segment, and the estimators' hand-over
// copy: one tiny launch instead of two pinned-memory uploads, an event and a device-to-device copy in front of the FFTs.
constexpr int kArgMax = 64;
struct notch_run_args { int n_ifirst, ndet; int ifirst[kArgMax + 1]; unsigned long long offs[kArgMax]; };
__global__ __launch_bounds__(64) void k_notch_args(notch_run_args r, int *d_ifirst, unsigned long long *d_offsets, const notch_est *carry_cur,
                                                   notch_est *carry_next) {
  const int t = threadIdx.x;
  for (int i = t; i < r.n_ifirst; i += 64) d_ifirst[i] = r.ifirst[i];
  for (int i = t; i < r.ndet; i += 64) d_offsets[i] = r.offs[i];
  if (t == 0 && carry_cur) *carry_next = *carry_cur;
}

// peak search of detect() (sdr.h:94-117) on one spectrum per workgroup: amplitudes by hypotf, nslots rounds of
// "first maximum wins, zero it and its two neighbours"
// The detect transforms of the scan mode, small enough to run NEXT TO two fir_filter workgroups on a CU (those hold 126 of the
// 160 KB of LDS and two thirds of the registers; a 4096-point transform in one workgroup — 32 KB of LDS — was seen in the kernel
// trace starting with a fir_filter launch and ending with it, 90-110 us for 15 us of work).  Stages 0..10 of cfft_engine's loop
// (dsp.h:84-104) touch the two halves of the bit-reversed array independently: one 256-lane workgroup with 16 KB of LDS per
// half; the last stage (position p with p + 2048, twiddle om[p]) is done by k_notch_peaks on the fly.  Same butterflies, each
// evaluated once with the same expression: bit-identical to k_cfft.
__device__ __forceinline__ void cfft_half_body(const float2 *src, const float2 *om, float2 *halves /*[ndet][2][2048]*/) {
  cfft_half_body_t([&](unsigned i) { return src[i]; }, om, halves, blockIdx.x);     // notch_detect.h
}
__global__ __launch_bounds__(256) void k_cfft_half(const float2 *in, const float2 *om, float2 *halves /*[ndet][2][2048]*/,
                                                   const unsigned long long *in_offsets) {
  cfft_half_body(in + in_offsets[blockIdx.x >> 1], om, halves);
}

__global__ __launch_bounds__(256) void k_notch_peaks(const float2 *halves, const float2 *om, float invn, int nslots,
                                                     int *cand /*[ndet][kMaxSlots]*/) {
  notch_peaks_body(halves, om, invn, nslots, cand, blockIdx.x);                      // notch_detect.h
}

// Interval q+1 of a run starts at detect point q with the slots' new bins (sdr.h:94-118): slot s of interval q uses
// cand[q−1][s] (interval 0: the carried bin), and its estimator restarts where that differs from the interval before.  Every
// (interval, slot) is independent, so the workgroup that builds the (q, s) phasor table works its own bin out — a separate
// single-thread planning kernel walked the detect points through dependent global loads and took 20 µs alone, 50 µs next to
// fir_filter.  The carried bins ping-pong (bins_cur is read by several workgroups, bins_next written by the last interval's).
// Phasor table: expj[i] = (cosf(a), sinf(a)), a = (float)(2π·bin·i/4096) (sdr.h:107-111); bin < 0 (nothing detected yet): zeros,
// like the reference's untouched slots.
__global__ __launch_bounds__(256) void k_notch_tables(const int *cand /*[ndet][kMaxSlots]*/, int ndet, int nslots, const int *bins_cur,
                                                      int *bins_next, unsigned char *reset /*[ndet+1][kMaxSlots]*/, float2 *tables) {
  const int q = blockIdx.x / nslots, s = blockIdx.x % nslots;
  const int bin = q == 0 ? bins_cur[s] : cand[(q - 1) * kMaxSlots + s];
  if (threadIdx.x == 0) {
    const int prev = q == 0 ? bin : (q == 1 ? bins_cur[s] : cand[(q - 2) * kMaxSlots + s]);
    reset[q * kMaxSlots + s] = (unsigned char)(bin != prev);
    if (q == ndet) bins_next[s] = bin;
  }
  if (s == 0 && (int)threadIdx.x >= nslots && (int)threadIdx.x < kMaxSlots) {   // unused slots: carried through, never reset
    reset[q * kMaxSlots + threadIdx.x] = 0;
    if (q == ndet) bins_next[threadIdx.x] = bins_cur[threadIdx.x];
  }
  float2 *tb = tables + ((size_t)q * nslots + s) * kN;
  for (int i = threadIdx.x; i < kN; i += 256) {
    if (bin < 0) { tb[i] = make_float2(0.f, 0.f); continue; }
    const float ang = (float)(2 * M_PI * bin * i / kN);
    tb[i] = make_float2(cosf(ang), sinf(ang));
  }
}


// ---------------------------------------------------------------- fused auto_notch (one slot) + fir_filter: "notch_fir"
// leandvb's DEFAULT graph puts auto_notch(1 slot) in front of fir_filter (leandvb.cc:103,296-301).  As separate blocks that is three
// kernels moving 24 B per input sample (scan 8 + 8, filter 8).  Fused, the notched stream never exists:
//   auto_notch::process (sdr.h:119-138), one slot, gain 1:  estim[n] = k·x[n]·conj(e[n]) + (1−k)·estim[n−1],  out[n] = x[n] − estim[n]·e[n],
//   e[n] = exp(jθn), θ = 2π·bin/4096.  With sub[n] = estim[n]·e[n]:  sub[n] = p·sub[n−1] + k·x[n],  p = (1−k)·exp(jθ)  —  the notch is the
//   LTI filter  H(z) = (1 − k − p·z⁻¹) / (1 − p·z⁻¹)  (zero ON the unit circle at θ, pole at p).
//   fir_filter (dsp.h:246-262) then takes  y[m] = Σ_i c[i]·out[N + m·D − i].  Since (1 − p^D z^−D) / (1 − p z⁻¹) = Σ_{j<D} p^j z^−j,
//       y[m] − P·y[m−1] = Σ_t ρ[t]·x[N + m·D − t],     P = p^D,   ρ = c ∗ κ,   κ = [1−k, −k·p, −k·p², …, −k·p^{D−1}, −p^D]   (N + D taps),
//   i.e. ONE decimating FIR with complex taps over the RAW samples (k_fir_mfma_stream<30, 1, 12, IV>: the matrix pipe, 8 B per sample)
//   and a first-order recurrence at the DECIMATED rate (k_nf_scan: |P| = 0.94, a 512-output warm-up forgets the start to 4·10⁻¹⁴).
// detect() (sdr.h:76-118) stays what it is — FFT of the detect block's input, first maximum — on the device (k_cfft_half, k_notch_peaks);
// the taps of every detect interval of a run are built on the device from its bin (k_nf_taps), the filter pass picks them by tile.
// Where a detect CHANGES the bin (estim ← 0, new e[]) the recurrence above does not hold for the outputs whose windows straddle the
// change: those few outputs (and the rest of the tile that still ran with the old taps) are computed directly — notch recurrences over
// a few thousand samples in one workgroup, then the filter sums (k_nf_fix) — and enter the scan as given values.
// State between runs: the last output, the last D raw samples before the read pointer (the ρ window reaches D samples further back than
// fir_filter's), the bin, and sub at the notch's frontier (so that a later bin change can reconstruct the old segment's estimator).
// Arithmetic: float32 with exact phases — NOT the reference's rounding sequence: a tolerance mode (include/lsdr_hip.h states the bound).
constexpr int kNfD = 30, kNfNq = 12, kNfKx = 8, kNfKs = 3 * kNfKx /* operand rows: re, −im, +im parts of 8 steps */, kNfTaps = kNfD * kNfNq;   // 360 tap slots, N + D ≤ 360
constexpr int kNfLook = 12288;            // samples an estimator remembers ((1−k)^12288 < 1e-8 is checked at run time, as in the scan mode)
constexpr int kNfMaxDet = 64;             // detect points per run (the run is cut there)
constexpr int kNfFixSpan = 4736;          // samples one fix-up stages (N + a tile and the transition outputs' windows)
constexpr int kNfChunk = 4096, kNfWarm = 512, kNfPer = (kNfChunk + kNfWarm) / 256;   // k_nf_scan geometry (18 elements per lane)

struct nf_state {
  float2 y_last;          // output M0−1
  float2 sub;             // sub[A−1] of the current segment (0 before the first detect)
  int bin, pad;           // −1: no detect yet
  float2 carry[32];       // x[F−32 … F): the raw samples before the read pointer
};

struct nf_run {           // geometry of one run (kernel-argument segment)
  int ndet;
  unsigned mw;                                  // outputs per wave tile of the filter pass
  unsigned long long count;                     // outputs of this run
  unsigned long long a_prev_rel, a_rel;         // notch frontier before / after the run, relative to `in`
  unsigned long long s_rel[kNfMaxDet];          // detect point q: first sample of its block, relative to `in`
  unsigned m_lo[kNfMaxDet], m_hi[kNfMaxDet];    // outputs given directly if the bin changes there (run-relative, inclusive; lo > hi: none)
  unsigned tile_first[kNfMaxDet + 1];           // interval q serves the filter tiles from tile_first[q] on ([0] = 0)
};

struct nf_consts { float k, omk, scale; int N; };

__device__ __forceinline__ float2 nf_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// p^e, p = omk·exp(j·2π·bin/4096): the phase by integer arithmetic (exact), the magnitude in double
__device__ __forceinline__ float2 nf_ppow(int bin, float omk, long long e) {
  const double mag = exp((double)e * log((double)omk));
  const double ang = 2.0 * M_PI * (double)(((long long)bin * e) & 4095) / 4096.0;
  return make_float2((float)(mag * cos(ang)), (float)(mag * sin(ang)));
}
// Σ_{n=lo}^{hi−1} p^{hi−1−n}·x[n] by a 256-thread workgroup (every thread gets the sum); x[n] = 0 for n < 0
__device__ float2 nf_wsum(const float2 *x, long long lo, long long hi, int bin, float omk, float2 *sh /*[256]*/) {
  const int t = threadIdx.x;
  const long long L = hi > lo ? hi - lo : 0, per = (L + 255) / 256;
  const long long b = lo + t * per, e = b + per < hi ? b + per : hi;
  float2 acc = make_float2(0.f, 0.f);
  if (b < e) {
    const float2 p = nf_ppow(bin, omk, 1);
    for (long long n = b; n < e; ++n) {
      const float2 v = n >= 0 ? x[n] : make_float2(0.f, 0.f);
      acc = nf_cmul(acc, p);
      acc.x += v.x; acc.y += v.y;
    }
    acc = nf_cmul(acc, nf_ppow(bin, omk, hi - e));
  }
  __syncthreads();
  sh[t] = acc;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (t < d) { sh[t].x += sh[t + d].x; sh[t].y += sh[t + d].y; }
    __syncthreads();
  }
  const float2 r = sh[0];
  __syncthreads();
  return r;
}

// the detect points' half transforms, offsets straight from the run's record
__global__ __launch_bounds__(256) void k_nf_cfft_half(nf_run r, const float2 *in, const float2 *om, float2 *halves) {
  cfft_half_body(in + r.s_rel[blockIdx.x >> 1], om, halves);
}

// Interval q of a run (q = 0: up to the first detect point of the run, carried bin; q ≥ 1: from detect point q−1 on): its bin, whether
// it differs from the interval before, P = p^D and the taps ρ = (scale·c) ∗ κ — in natural order (k_nf_scan computes r[0] with them) and as the filter pass's
// coefficient operand (lane (k = l>>4, q' = l&15) of step s: tap D·q' + 4·s + k of tap block q' — its re part in table row s, −im in row 8 + s, +im in row 16 + s: fir_stream.h).
// The carried bin travels from one run's launch of this kernel to the next one's through a ping-pong word (bin_in / bin_out), not through
// nf_state: with lsdr_notch_fir_set_overlap the detect chain of run k+1 runs while run k's tail still owns the state.
__global__ __launch_bounds__(256) void k_nf_taps(nf_run run, const int *bin_in, int *bin_out, const int *cand /*[ndet][kMaxSlots]*/, const float2 *coeffs,
                                                 nf_consts C, int *ivbin, unsigned char *changed, float2 *ivP, float2 *ivrho /*[·][kNfTaps]*/,
                                                 float *ivtab /*[·][kNfKs·64]*/, unsigned *tile_first) {
  const int ndet = run.ndet;
  if (blockIdx.x == 0) for (int i = threadIdx.x; i <= ndet; i += 256) tile_first[i] = run.tile_first[i];     // (the filter pass reads them through a pointer)
  __shared__ double kr[kNfD + 1], ki[kNfD + 1];
  __shared__ double rr[kNfTaps], ri[kNfTaps];
  __shared__ float2 cs_s[kNfTaps];                 // the filter's taps (a per-tap global load inside the convolution made this kernel 26 µs)
  const int q = blockIdx.x, t = threadIdx.x;
  for (int i = t; i < kNfTaps; i += 256) cs_s[i] = i < C.N ? coeffs[i] : make_float2(0.f, 0.f);
  const int bin = q == 0 ? *bin_in : cand[(q - 1) * kMaxSlots];
  const int prev = q == 0 ? bin : (q == 1 ? *bin_in : cand[(q - 2) * kMaxSlots]);
  if (t == 0 && q == ndet) *bin_out = bin;
  if (t <= kNfD) {
    double re = 0, im = 0;
    if (bin < 0) { re = t == 0 ? 1.0 : 0.0; }
    else if (t == 0) { re = 1.0 - (double)C.k; }
    else {
      const double mag = exp((double)t * log((double)C.omk)), ang = 2.0 * M_PI * (double)((bin * t) & 4095) / 4096.0;
      const double g = t < kNfD ? -(double)C.k : -1.0;
      re = g * mag * cos(ang); im = g * mag * sin(ang);
    }
    kr[t] = re; ki[t] = im;
  }
  __syncthreads();
  if (t == 0) {
    ivbin[q] = bin; changed[q] = (unsigned char)(q > 0 && bin != prev);
    ivP[q] = bin < 0 ? make_float2(0.f, 0.f) : make_float2((float)-kr[kNfD], (float)-ki[kNfD]);
  }
  for (int tt = t; tt < kNfTaps; tt += 256) {
    double re = 0, im = 0;
    for (int j = 0; j <= kNfD; ++j) {
      const int i = tt - j;
      if (i < 0 || i >= C.N) continue;
      const float2 cs = cs_s[i];        // fir_filter's shifted taps (dsp.h:271-280) with the fused scaler on them: one f32 rounding per component
      re += kr[j] * (double)cs.x - ki[j] * (double)cs.y; im += kr[j] * (double)cs.y + ki[j] * (double)cs.x;
    }
    rr[tt] = re; ri[tt] = im;
    ivrho[(size_t)q * kNfTaps + tt] = make_float2((float)re, (float)im);
  }
  __syncthreads();
  for (int idx = t; idx < kNfKs * 64; idx += 256) {
    const int row = idx >> 6, part = row / kNfKx, s = row - part * kNfKx, ln = idx & 63, r = 4 * s + (ln >> 4), qq = ln & 15;
    float v = 0.f;
    if (qq < kNfNq && r < kNfD) v = part == 0 ? (float)rr[kNfD * qq + r] : part == 1 ? (float)-ri[kNfD * qq + r] : (float)ri[kNfD * qq + r];
    ivtab[(size_t)q * (kNfKs * 64) + idx] = v;
  }
}

// which boundary before `q` (exclusive) last changed the bin in this run: its index, or −1 (the segment came in with the run)
__device__ __forceinline__ int nf_seg_start(const unsigned char *changed, int q) {
  for (int i = q - 1; i >= 1; --i) if (changed[i]) return i;
  return -1;
}

// Detect point q (interval q+1 begins at sample S) changed the bin: the outputs [m_lo, m_hi] directly.
__global__ __launch_bounds__(256) void k_nf_fix(nf_run run, const float2 *in, const nf_state *st, const int *ivbin, const unsigned char *changed,
                                                const float2 *coeffs, nf_consts C, float2 *r) {
  __shared__ float2 xs[kNfFixSpan];
  __shared__ float2 sh[256];
  const int q = blockIdx.x, t = threadIdx.x;
  if (!changed[q + 1]) return;
  const unsigned mlo = run.m_lo[q], mhi = run.m_hi[q];
  if (mlo > mhi) return;
  const long long S = (long long)run.s_rel[q];
  const int N = C.N, bin_old = ivbin[q], bin_new = ivbin[q + 1];
  const long long n_first = S - N, n_last = (long long)N + (long long)mhi * kNfD;
  const int span = (int)(n_last - n_first + 1);          // ≤ kNfFixSpan by the run's geometry (host)
  for (int i = t; i < span; i += 256) {
    const long long n = n_first + i;
    xs[i] = n >= 0 ? in[n] : make_float2(0.f, 0.f);
  }
  // the old segment's sub[S−1]
  float2 sub_old = make_float2(0.f, 0.f);
  if (bin_old >= 0) {
    const int sg = nf_seg_start(changed, q + 1);
    const bool carried = sg < 0;
    const long long start = carried ? (long long)run.a_prev_rel : (long long)run.s_rel[sg - 1];
    const long long lo = S - kNfLook > start ? S - kNfLook : start;
    const float2 w = nf_wsum(in, lo, S, bin_old, C.omk, sh);
    sub_old = make_float2(C.k * w.x, C.k * w.y);
    if (carried && S - start <= kNfLook) {
      const float2 c2 = nf_cmul(nf_ppow(bin_old, C.omk, S - start), st->sub);
      sub_old.x += c2.x; sub_old.y += c2.y;
    }
  }
  __syncthreads();
  if (t == 0) {            // the two recurrences, sequentially (a few thousand steps, once per bin change)
    const int iS = (int)(S - n_first);
    if (bin_old >= 0) {
      const float2 p = nf_ppow(bin_old, C.omk, 1);
      const float inv = 1.0f / (p.x * p.x + p.y * p.y);
      const float2 ip = make_float2(p.x * inv, -p.y * inv);
      float2 s = sub_old;
      for (int i = iS - 1; i >= 0; --i) {
        const float2 x = xs[i];
        xs[i] = make_float2(x.x - s.x, x.y - s.y);
        s = nf_cmul(make_float2(s.x - C.k * x.x, s.y - C.k * x.y), ip);
      }
    }
    {
      const float2 p = nf_ppow(bin_new, C.omk, 1);
      float2 s = make_float2(0.f, 0.f);
      for (int i = iS; i < span; ++i) {
        const float2 x = xs[i];
        s = nf_cmul(p, s);
        s.x += C.k * x.x; s.y += C.k * x.y;
        xs[i] = make_float2(x.x - s.x, x.y - s.y);
      }
    }
  }
  __syncthreads();
  for (unsigned m = mlo + t; m <= mhi; m += 256) {
    const int top = (int)((long long)N + (long long)m * kNfD - n_first);
    float2 acc = make_float2(0.f, 0.f);
    for (int i = 0; i < N; ++i) {
      const float2 cs = coeffs[i], v = xs[top - i];
      acc.x = fmaf(cs.x, v.x, acc.x); acc.x = fmaf(-cs.y, v.y, acc.x);
      acc.y = fmaf(cs.x, v.y, acc.y); acc.y = fmaf(cs.y, v.x, acc.y);
    }
    r[m] = acc;
  }
}

// fir_filter re-shifted its taps between two runs (dsp.h:236-244: the receiver's carrier estimate moved them): the recurrence wants the
// output BEFORE this run's first one as the NEW taps would have given it — Σ_i c'[i]·out[N − D − i] over the notched samples, which reach
// D − 1 samples back past `in` (the carried ones) and whose estimator is stepped BACKWARDS from its carried value at the notch's frontier
// (sub[n−1] = (sub[n] − k·x[n]) / p; the frontier is 313–342 samples past `in`, the segment began at least a block earlier).  One workgroup.
__global__ __launch_bounds__(256) void k_nf_retap(nf_run run, const float2 *in, nf_state *st, const float2 *coeffs, nf_consts C) {
  __shared__ float2 xs[512];
  __shared__ float2 sh[256];
  const int t = threadIdx.x, N = C.N;
  const int Ap = (int)run.a_prev_rel;                 // sub is known at sample Ap − 1
  const int lo = -(kNfD - 1), span = Ap - lo;         // samples lo … Ap − 1
  for (int i = t; i < span; i += 256) { const int n = lo + i; xs[i] = n >= 0 ? in[n] : st->carry[32 + n]; }
  __syncthreads();
  if (t == 0 && st->bin >= 0) {
    const float2 p = nf_ppow(st->bin, C.omk, 1);
    const float inv = 1.0f / (p.x * p.x + p.y * p.y);
    const float2 ip = make_float2(p.x * inv, -p.y * inv);
    float2 s = st->sub;
    for (int i = span - 1; i >= 0; --i) {
      const float2 x = xs[i];
      xs[i] = make_float2(x.x - s.x, x.y - s.y);
      s = nf_cmul(make_float2(s.x - C.k * x.x, s.y - C.k * x.y), ip);
    }
  }
  __syncthreads();
  float2 acc = make_float2(0.f, 0.f);
  for (int i = t; i < N; i += 256) {
    const float2 v = nf_cmul(coeffs[i], xs[(N - kNfD - i) - lo]);
    acc.x += v.x; acc.y += v.y;
  }
  sh[t] = acc;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (t < d) { sh[t].x += sh[t + d].x; sh[t].y += sh[t + d].y; }
    __syncthreads();
  }
  if (t == 0) st->y_last = sh[0];
}

// y[m] = A_m·y[m−1] + r[m] over the run: A_m = P of the interval that served output m's filter tile, 0 where r[m] is a given value.
// One workgroup per kNfChunk outputs, started kNfWarm outputs early from zero (block 0: from the carried output, exactly).  The chunk
// goes through LDS both ways (coalesced 8-byte accesses; a lane's kNfPer consecutive elements sit kNfPer + 1 slots apart: no conflicts
// worth the name) — with 80-byte strides between lanes the kernel ran at 1.9 TB/s.
// Block 0 also computes r[0] — its window reaches D − 1 samples back past `in` (the carried ones); the filter pass read zeros there —
// and the block that holds the run's last output leaves what the next run needs in `so` (the OTHER state record: block 0 of this launch
// may still be reading `st`): the last output, the raw samples in front of the new read pointer, the bin, sub at the new frontier.
struct nf_aff { float2 a, b; };     // y → a·y + b
__device__ __forceinline__ nf_aff nf_then(const nf_aff &f, const nf_aff &g) {   // f first, then g
  nf_aff o; o.a = nf_cmul(g.a, f.a); o.b = nf_cmul(g.a, f.b); o.b.x += g.b.x; o.b.y += g.b.y; return o;
}
__device__ __forceinline__ void nf_state_body(const nf_run &run, const float2 *in, const nf_state *st, nf_state *so, const int *ivbin,
                                              const unsigned char *changed, const nf_consts &C, float2 *sh);
constexpr int kNfSlots = (kNfChunk + kNfWarm) + (kNfChunk + kNfWarm) / kNfPer;
__device__ __forceinline__ int nf_slot(int i) { return i + i / kNfPer; }
__global__ __launch_bounds__(256) void k_nf_scan(nf_run run, const float2 *in, const float2 *r, const nf_state *st, nf_state *so, const float2 *ivP,
                                                 const float2 *ivrho, const int *ivbin, const unsigned char *changed, nf_consts C, float2 *out) {
  __shared__ float2 rs[kNfSlots];
  __shared__ nf_aff sc[2][256];
  __shared__ float2 sh[256];
  __shared__ unsigned long long s_mask;
  const int t = threadIdx.x;
  if (blockIdx.x == gridDim.x - 1) { nf_state_body(run, in, st, so, ivbin, changed, C, sh); return; }      // (the grid's extra block)
  const long long c0 = (long long)blockIdx.x * kNfChunk, base = c0 - kNfWarm, cnt = (long long)run.count;
  const long long hi = c0 + kNfChunk < cnt ? c0 + kNfChunk : cnt;
  if (t == 0) s_mask = 0ull;
  for (int i = t; i < kNfChunk + kNfWarm; i += 256) {
    const long long m = base + i;
    rs[nf_slot(i)] = (m >= 0 && m < hi) ? r[m] : make_float2(0.f, 0.f);
  }
  __syncthreads();
  if (t < run.ndet && changed[t + 1] && run.m_lo[t] <= run.m_hi[t] && (long long)run.m_hi[t] >= base && (long long)run.m_lo[t] < hi)
    atomicOr(&s_mask, 1ull << t);
  if (blockIdx.x == 0) {          // r[0]
    float2 acc = make_float2(0.f, 0.f);
    for (int tt = t; tt < C.N + kNfD; tt += 256) {
      const int i = C.N - tt;
      const float2 x = i >= 0 ? in[i] : st->carry[32 + i];
      const float2 v = nf_cmul(ivrho[tt], x);
      acc.x += v.x; acc.y += v.y;
    }
    sh[t] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
      if (t < d) { sh[t].x += sh[t + d].x; sh[t].y += sh[t + d].y; }
      __syncthreads();
    }
  }
  __syncthreads();
  const unsigned long long mask = s_mask;
  if (blockIdx.x == 0 && t == 0) {
    bool given0 = false;
    for (int q = 0; q < run.ndet; ++q) if ((mask >> q & 1ull) && run.m_lo[q] == 0u) given0 = true;
    if (!given0) rs[nf_slot(kNfWarm)] = sh[0];
  }
  __syncthreads();
  const int e0 = t * kNfPer;
  const long long i0 = base + e0;
  int iv = 0;
  {
    const long long mf = i0 > 0 ? i0 : 0;
    const unsigned lt = (unsigned)(mf / run.mw);
    while (iv < run.ndet && lt >= run.tile_first[iv + 1]) ++iv;
  }
  float2 A[kNfPer], B[kNfPer];
  nf_aff loc; loc.a = make_float2(1.f, 0.f); loc.b = make_float2(0.f, 0.f);
#pragma unroll
  for (int e = 0; e < kNfPer; ++e) {
    const long long m = i0 + e;
    float2 a = make_float2(1.f, 0.f), b = make_float2(0.f, 0.f);
    if (m >= 0 && m < hi) {
      const unsigned lt = (unsigned)(m / run.mw);
      while (iv < run.ndet && lt >= run.tile_first[iv + 1]) ++iv;
      a = ivP[iv];
      b = rs[e0 + e + t];          // = nf_slot(e0 + e)
      if (mask) {
        for (int q = 0; q < run.ndet; ++q)
          if ((mask >> q & 1ull) && m >= (long long)run.m_lo[q] && m <= (long long)run.m_hi[q]) a = make_float2(0.f, 0.f);
      }
    }
    A[e] = a; B[e] = b;
    nf_aff g; g.a = a; g.b = b;
    loc = nf_then(loc, g);
  }
  sc[0][t] = loc;
  __syncthreads();
  int cur = 0;
  for (int d = 1; d < 256; d <<= 1) {
    nf_aff v = sc[cur][t];
    if (t >= d) v = nf_then(sc[cur][t - d], v);
    sc[cur ^ 1][t] = v;
    cur ^= 1;
    __syncthreads();
  }
  float2 y = blockIdx.x == 0 ? st->y_last : make_float2(0.f, 0.f);     // value in front of the block's first element
  if (t > 0) {
    const nf_aff ex = sc[cur][t - 1];
    const float2 ay = nf_cmul(ex.a, y);
    y = make_float2(ay.x + ex.b.x, ay.y + ex.b.y);
  }
#pragma unroll
  for (int e = 0; e < kNfPer; ++e) {
    const float2 ay = nf_cmul(A[e], y);
    y = make_float2(ay.x + B[e].x, ay.y + B[e].y);
    rs[e0 + e + t] = y;
  }
  __syncthreads();
  for (int i = kNfWarm + t; i < kNfChunk + kNfWarm; i += 256) {
    const long long m = base + i;
    if (m < hi) out[m] = rs[nf_slot(i)];
  }
  if (hi == cnt && t == 0) so->y_last = rs[nf_slot((int)(cnt - 1 - base))];
}

// the rest of what the next run needs — the raw samples in front of the new read pointer, the bin, sub at the new frontier — into `so`;
// launched as one more block of k_nf_scan's grid would have made that launch as long as scan + this (80 µs): its own workgroup in the same
// launch instead (k_nf_scan_state below)
__device__ __forceinline__ void nf_state_body(const nf_run &run, const float2 *in, const nf_state *st, nf_state *so, const int *ivbin,
                                              const unsigned char *changed, const nf_consts &C, float2 *sh) {
  const int t = threadIdx.x;
  const long long cnt = (long long)run.count;
  const int bin = ivbin[run.ndet];
  const long long Af = (long long)run.a_rel, Ap = (long long)run.a_prev_rel;
  float2 sub = make_float2(0.f, 0.f);
  if (bin >= 0) {
    const int sg = nf_seg_start(changed, run.ndet + 1);
    const bool carried = sg < 0;
    const long long start = carried ? Ap : (long long)run.s_rel[sg - 1];
    const long long lo = Af - kNfLook > start ? Af - kNfLook : start;
    const float2 w = nf_wsum(in, lo, Af, bin, C.omk, sh);
    sub = make_float2(C.k * w.x, C.k * w.y);
    if (carried && Af - start <= kNfLook) {
      const float2 c2 = nf_cmul(nf_ppow(bin, C.omk, Af - start), st->sub);
      sub.x += c2.x; sub.y += c2.y;
    }
  }
  const long long F = cnt * kNfD;                            // the new read pointer, relative to `in`
  if (t < 32) { const long long n = F - 32 + t; so->carry[t] = n >= 0 ? in[n] : st->carry[32 + n]; }   // (n ≥ −32: the old carry, shifted)
  if (t == 0) { so->sub = sub; so->bin = bin; so->pad = 0; }
}

}  // namespace

struct lsdr_auto_notch {
  lsdr_ctx *ctx;
  int nslots, decimation, phase;
  float k, gain, agc_rms_setpoint;
  int bins[kMaxSlots];
  bool any_active;
  std::vector<lsdr_cf32> expj;     // [nslots][4096] host copy
  float2 *d_expj;
  notch_est est;                   // carried estimators (host mirror)
  notch_est *d_carry, *d_begin, *d_end;
  size_t tiles_cap;
  unsigned last_tiles, last_bad;
  cfft_dev fft;
  // throughput mode (LSDR_NOTCH_SCAN): everything below lives on the device between runs
  int mode;
  bool scan_started;
  notch_est *d_scarry[2]; int scarry_cur;     // estimators, ping-pong (a run reads one and writes the other)
  int *d_bins;                                // [2][kMaxSlots] carried bins, ping-pong with d_scarry
  unsigned long long *d_offsets; float2 *d_spec; int *d_cand; int *d_ibins; unsigned char *d_reset; int *d_ifirst;
  size_t det_cap;                             // detect points the scratch above is sized for
  float2 *d_tables; size_t tables_cap;        // [(ndet+1)·nslots·4096]
  float2 *d_totals; unsigned *d_flags; size_t blocks_cap;
  unsigned *h_abort, *d_abort;               // pinned word (host / device view): a look-back wait of k_notch_scan gave up
  // lsdr_auto_notch_set_overlap: the detect chain of run k+1 (FFTs → peaks → tables: it depends on the INPUT and on the bins of
  // chain k only) on a side stream, next to k_notch_scan of run k; two sets of the chain's buffers, used alternately
  bool overlap; hipStream_t side; hipEvent_t ev_chain[2], ev_scan[2]; unsigned run_no;
  // optional timing of the scan kernel alone (lsdr_auto_notch_scan_time): a ring of event pairs around its launches
  static const int kTimed = 16;
  bool timing;
  hipEvent_t tev[kTimed][2];
  unsigned timed_runs;
  unsigned stamp;
  // per-run argument arrays (interval starts, detect offsets) cross over from two pinned slots used alternately: the copy of
  // run k-2 has long executed when its slot is rewritten, so the host never waits for the GPU in steady state
  int *h_ifirst[2]; unsigned long long *h_offsets[2]; size_t h_cap[2]; hipEvent_t h_ev[2]; int h_slot;
};

struct lsdr_spectrum {
  lsdr_ctx *ctx;
  int decimation, phase;
  float kavg;
  std::vector<float> avgpower;   // empty until the first spectrum
  cfft_dev fft;
};

struct lsdr_cnr_fft {
  lsdr_ctx *ctx;
  float bandwidth, kavg;
  int nfft, decimation, phase;
  std::vector<float> avgpower;     // empty until the first spectrum
  cfft_dev fft;
};

// detect(), sdr.h:76-118: AGC sums on a host copy of the block, peak search on its spectrum (FFT done by k_cfft)
static void notch_detect(lsdr_auto_notch *a, const lsdr_cf32 *pin, const std::vector<lsdr_cf32> &data) {
  float m0 = 0, m2 = 0;
  for (int i = 0; i < kN; ++i) {
    m2 += (float)pin[i].re * pin[i].re + (float)pin[i].im * pin[i].im;
    if (fabsf(pin[i].re) > m0) m0 = fabsf(pin[i].re);
    if (fabsf(pin[i].im) > m0) m0 = fabsf(pin[i].im);
  }
  if (a->agc_rms_setpoint && m2) {
    float rms = sqrtf(m2 / kN);
    float new_gain = a->agc_rms_setpoint / rms;
    a->gain = (float)((double)a->gain * 0.9 + (double)new_gain * 0.1);
  }
  std::vector<float> amp(kN);
  for (int i = 0; i < kN; ++i) amp[i] = hypotf(data[i].re, data[i].im);
  for (int s = 0; s < a->nslots; ++s) {
    int iamax = 0;
    for (int i = 0; i < kN; ++i) if (amp[i] > amp[iamax]) iamax = i;
    if (iamax != a->bins[s]) {
      a->bins[s] = iamax;
      a->est.re[s] = 0; a->est.im[s] = 0;
      for (int i = 0; i < kN; ++i) {
        float ang = (float)(2 * M_PI * iamax * i / kN);
        a->expj[(size_t)s * kN + i].re = cosf(ang);
        a->expj[(size_t)s * kN + i].im = sinf(ang);
      }
    }
    amp[iamax] = 0;
    if (iamax - 1 >= 0) amp[iamax - 1] = 0;
    if (iamax + 1 < kN) amp[iamax + 1] = 0;
  }
  a->any_active = a->nslots > 0;
}

template <int NS>
static void notch_launch_ns(hipStream_t st, unsigned grid, const notch_args &a) {
  const size_t lds = NS <= kLdsSlots ? (size_t)NS * kN * sizeof(float2) : 0;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_notch<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_notch<NS>, dim3(grid), dim3(64), lds, st, a);
}
static void notch_launch(int ns, hipStream_t st, unsigned grid, const notch_args &a) {
  switch (ns) {
    case 1: notch_launch_ns<1>(st, grid, a); break;
    case 2: notch_launch_ns<2>(st, grid, a); break;
    case 3: notch_launch_ns<3>(st, grid, a); break;
    case 4: notch_launch_ns<4>(st, grid, a); break;
    case 5: notch_launch_ns<5>(st, grid, a); break;
    case 6: notch_launch_ns<6>(st, grid, a); break;
    case 7: notch_launch_ns<7>(st, grid, a); break;
    default: notch_launch_ns<8>(st, grid, a); break;
  }
}

// process() over `nb` blocks with fixed slots: tiled + verified
static int notch_process(lsdr_auto_notch *a, const lsdr_cf32 *in, lsdr_cf32 *out, size_t nb) {
  lsdr_ctx *c = a->ctx;
  if (!nb) return LSDR_OK;
  if (!a->any_active) {
    size_t n = nb * kN;
    size_t blocks = (n + 255) / 256, capb = (size_t)c->num_cu * 8;
    if (blocks > capb) blocks = capb;
    hipLaunchKernelGGL(k_notch_passthrough, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const float2 *)in, (float2 *)out, n, a->gain);
    LSDR_HIP(hipGetLastError());
    return LSDR_OK;
  }
  // tiles: tile 0 covers the first (kWarmBlocks + kTileBlocks) blocks so that every other tile can warm up fully
  const unsigned TB = kTileBlocks, WB = kWarmBlocks;
  auto tile_start = [&](unsigned t) -> unsigned long long { return t == 0 ? 0ull : (unsigned long long)WB + TB + (unsigned long long)(t - 1) * TB; };
  unsigned n_tiles = nb > (size_t)(WB + TB) ? 1u + (unsigned)((nb - (WB + TB) + TB - 1) / TB) : 1u;
  if (a->tiles_cap < n_tiles) {
    (void)hipFree(a->d_begin); (void)hipFree(a->d_end);
    LSDR_HIP(hipMalloc((void **)&a->d_begin, n_tiles * sizeof(notch_est)));
    LSDR_HIP(hipMalloc((void **)&a->d_end, n_tiles * sizeof(notch_est)));
    a->tiles_cap = n_tiles;
  }
  LSDR_HIP(hipMemcpyAsync(a->d_carry, &a->est, sizeof(notch_est), hipMemcpyHostToDevice, c->stream));
  LSDR_HIP(hipMemcpyAsync(a->d_expj, a->expj.data(), (size_t)a->nslots * kN * sizeof(float2), hipMemcpyHostToDevice, c->stream));
  notch_args na;
  na.in = (const float2 *)in; na.out = (float2 *)out; na.expj = a->d_expj; na.nslots = a->nslots;
  na.k = a->k; na.gain = a->gain; na.n_blocks = nb; na.tile_blocks = TB; na.warm_blocks = WB; na.n_tiles = n_tiles;
  na.carry = a->d_carry; na.begin = a->d_begin; na.end = a->d_end; na.serial_from = -1;
  notch_launch(a->nslots, c->stream, (n_tiles + kNotchLanes - 1) / kNotchLanes, na);
  LSDR_HIP(hipGetLastError());
  std::vector<notch_est> hb(n_tiles), he(n_tiles);
  LSDR_TRY(lsdr_stage_d2h(c, hb.data(), a->d_begin, n_tiles * sizeof(notch_est)));
  LSDR_TRY(lsdr_stage_d2h(c, he.data(), a->d_end, n_tiles * sizeof(notch_est)));
  LSDR_TRY(lsdr_stage_sync(c));
  a->last_tiles += n_tiles;
  // seam verification
  unsigned first_bad = n_tiles;
  for (unsigned t = 1; t < n_tiles; ++t) {
    bool ok = true;
    for (int s = 0; ok && s < a->nslots; ++s)
      ok = memcmp(&hb[t].re[s], &he[t - 1].re[s], 4) == 0 && memcmp(&hb[t].im[s], &he[t - 1].im[s], 4) == 0;
    if (!ok) { first_bad = t; break; }
  }
  if (first_bad < n_tiles) {
    // Re-verify after a sequential repair of the unverified head only: redo blocks [first_bad·TB, next verified seam)
    // sequentially from tile first_bad−1's end state, and continue until a later seam checks out.
    unsigned t = first_bad;
    notch_est state = he[t - 1];
    while (t < n_tiles) {
      // find how far the sequential pass must go: until tile u (> t) whose begin state equals our running end state.
      // Simple and exact: redo ONE tile sequentially, compare with the next tile's begin state, repeat.
      ++a->last_bad;
      LSDR_HIP(hipMemcpyAsync(a->d_carry, &state, sizeof(notch_est), hipMemcpyHostToDevice, c->stream));
      notch_args ra = na;
      ra.serial_from = (int)tile_start(t);
      unsigned long long bend = tile_start(t + 1);
      if (bend > nb) bend = nb;
      ra.n_blocks = bend;
      ra.n_tiles = 1;
      notch_launch(a->nslots, c->stream, 1, ra);
      LSDR_HIP(hipGetLastError());
      LSDR_HIP(hipMemcpyAsync(&state, a->d_end, sizeof(notch_est), hipMemcpyDeviceToHost, c->stream));
      LSDR_HIP(hipStreamSynchronize(c->stream));
      he[t] = state;
      ++t;
      if (t < n_tiles) {
        bool ok = true;
        for (int s = 0; ok && s < a->nslots; ++s)
          ok = memcmp(&hb[t].re[s], &state.re[s], 4) == 0 && memcmp(&hb[t].im[s], &state.im[s], 4) == 0;
        if (ok) {   // the speculative tiles from here on started from the right state; keep scanning their seams
          unsigned u = t + 1;
          for (; u < n_tiles; ++u) {
            bool ok2 = true;
            for (int s = 0; ok2 && s < a->nslots; ++s)
              ok2 = memcmp(&hb[u].re[s], &he[u - 1].re[s], 4) == 0 && memcmp(&hb[u].im[s], &he[u - 1].im[s], 4) == 0;
            if (!ok2) break;
          }
          if (u >= n_tiles) { t = n_tiles; break; }
          state = he[u - 1];
          t = u;
        }
      }
    }
  }
  a->est = he[n_tiles - 1];
  return LSDR_OK;
}

// LSDR_NOTCH_SCAN: one run = [batched detect FFTs → peaks → plan → tables] → one k_notch_scan launch, all enqueued, no
// host synchronisation (the caller's consumed/produced are pure functions of the sizes).
template <int NS>
static void notch_scan_launch(hipStream_t st, unsigned grid, const notch_scan_args &a) {
  hipLaunchKernelGGL(k_notch_scan<NS>, dim3(grid * kWavesPerBlock), dim3(64), 0, st, a);
}
static int notch_run_scan(lsdr_auto_notch *a, const lsdr_cf32 *in, lsdr_cf32 *out, size_t nb) {
  lsdr_ctx *c = a->ctx;
  {   // refused BEFORE anything of the block's state is touched
    const float a12288 = (float)pow(1.0 - (double)a->k, 12288);
    if (!(a12288 < 1e-8f)) {
      lsdr_set_error("auto_notch: LSDR_NOTCH_SCAN looks 12288 samples back; k=%g leaves (1-k)^12288=%g of older input (use LSDR_NOTCH_EXACT)", (double)a->k, (double)a12288);
      return LSDR_E_UNSUPPORTED;
    }
  }
  if (!a->scan_started) {
    for (int i = 0; i < 2; ++i) {
      LSDR_HIP(hipMalloc((void **)&a->d_scarry[i], sizeof(notch_est)));
      LSDR_HIP(hipMemcpyAsync(a->d_scarry[i], &a->est, sizeof(notch_est), hipMemcpyHostToDevice, c->stream));
    }
    LSDR_HIP(hipMalloc((void **)&a->d_bins, 2 * kMaxSlots * sizeof(int)));   // ping-pong like d_scarry (same index)
    for (int i = 0; i < 2; ++i) LSDR_HIP(hipMemcpyAsync(a->d_bins + i * kMaxSlots, a->bins, kMaxSlots * sizeof(int), hipMemcpyHostToDevice, c->stream));
    a->scarry_cur = 0; a->stamp = 0; a->scan_started = true;
  }
  // detect points of this run: `phase += 4096; if (phase >= decimation) { phase -= decimation; detect(); }` per block
  std::vector<int> ifirst(1, 0);
  std::vector<unsigned long long> offs;
  int phase = a->phase;
  for (size_t b = 0; b < nb; ++b) {
    phase += kN;
    if (phase >= a->decimation) { phase -= a->decimation; ifirst.push_back((int)b); offs.push_back((unsigned long long)b * kN); }
  }
  a->phase = phase;
  const size_t ndet = offs.size();
  const int ns = a->nslots;
  if (a->overlap && !a->side) {
    LSDR_HIP(hipStreamCreateWithFlags(&a->side, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      LSDR_HIP(hipEventCreateWithFlags(&a->ev_chain[i], hipEventDisableTiming));
      LSDR_HIP(hipEventCreateWithFlags(&a->ev_scan[i], hipEventDisableTiming));
    }
  }
  auto sync_all = [&]() -> int {
    LSDR_HIP(hipStreamSynchronize(c->stream));
    if (a->side) LSDR_HIP(hipStreamSynchronize(a->side));
    return LSDR_OK;
  };
  if (a->det_cap < ndet + 1) {     // (two sets of everything the detect chain writes: [set 0 | set 1])
    LSDR_TRY(sync_all());
    (void)hipFree(a->d_offsets); (void)hipFree(a->d_spec); (void)hipFree(a->d_cand); (void)hipFree(a->d_ibins); (void)hipFree(a->d_reset); (void)hipFree(a->d_ifirst);
    const size_t cap = ndet + 8;
    LSDR_HIP(hipMalloc((void **)&a->d_offsets, 2 * cap * sizeof(unsigned long long)));
    LSDR_HIP(hipMalloc((void **)&a->d_spec, 2 * cap * kN * sizeof(float2)));
    LSDR_HIP(hipMalloc((void **)&a->d_cand, 2 * cap * kMaxSlots * sizeof(int)));
    LSDR_HIP(hipMalloc((void **)&a->d_ibins, 2 * (cap + 1) * kMaxSlots * sizeof(int)));
    LSDR_HIP(hipMalloc((void **)&a->d_reset, 2 * (cap + 1) * kMaxSlots));
    LSDR_HIP(hipMalloc((void **)&a->d_ifirst, 2 * (cap + 1) * sizeof(int)));
    a->det_cap = cap;
  }
  if (a->tables_cap < (ndet + 1) * (size_t)ns * kN) {
    LSDR_TRY(sync_all());
    (void)hipFree(a->d_tables);
    a->tables_cap = (ndet + 8) * (size_t)ns * kN;
    LSDR_HIP(hipMalloc((void **)&a->d_tables, 2 * a->tables_cap * sizeof(float2)));
  }
  const unsigned par = a->overlap ? (a->run_no & 1u) : 0u;
  ++a->run_no;
  unsigned long long *const p_offsets = a->d_offsets + par * a->det_cap;
  float2 *const p_spec = a->d_spec + (size_t)par * a->det_cap * kN;
  int *const p_cand = a->d_cand + (size_t)par * a->det_cap * kMaxSlots;
  unsigned char *const p_reset = a->d_reset + (size_t)par * (a->det_cap + 1) * kMaxSlots;
  int *const p_ifirst = a->d_ifirst + (size_t)par * (a->det_cap + 1);
  float2 *const p_tables = a->d_tables + (size_t)par * a->tables_cap;
  hipStream_t cs = a->overlap ? a->side : c->stream;                 // the detect chain's stream
  if (a->overlap && a->run_no > 2) LSDR_HIP(hipStreamWaitEvent(cs, a->ev_scan[par], 0));   // the scan two runs ago read this set
  if (a->blocks_cap < nb) {
    LSDR_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(a->d_totals); (void)hipFree(a->d_flags);
    LSDR_HIP(hipMalloc((void **)&a->d_totals, (size_t)kMaxSlots * nb * kWavesPerBlock * sizeof(float2)));
    LSDR_HIP(hipMalloc((void **)&a->d_flags, (size_t)kMaxSlots * nb * kWavesPerBlock * sizeof(unsigned)));
    LSDR_HIP(hipMemsetAsync(a->d_flags, 0, (size_t)kMaxSlots * nb * kWavesPerBlock * sizeof(unsigned), c->stream));
    a->blocks_cap = nb;
    a->stamp = 0;
  }
  // estimators: the run reads d_scarry[cur] and writes the end state to d_scarry[nxt] — a copy made up front, so that the
  // first wave-blocks (which read the start state in their look-back) never see the last one's update
  const int cur = a->scarry_cur, nxt = cur ^ 1;
  if (ndet <= (size_t)kArgMax) {
    notch_run_args ra;
    ra.n_ifirst = (int)ifirst.size(); ra.ndet = (int)ndet;
    for (size_t i = 0; i < ifirst.size(); ++i) ra.ifirst[i] = ifirst[i];
    for (size_t i = 0; i < ndet; ++i) ra.offs[i] = offs[i];
    hipLaunchKernelGGL(k_notch_args, dim3(1), dim3(64), 0, cs, ra, p_ifirst, p_offsets,
                       a->overlap ? (const notch_est *)nullptr : (const notch_est *)a->d_scarry[cur], a->d_scarry[nxt]);
    // (overlapped: the estimators' hand-over copy reads what the PREVIOUS scan's last wave-block writes — main stream, below)
    if (a->overlap) LSDR_HIP(hipMemcpyAsync(a->d_scarry[nxt], a->d_scarry[cur], sizeof(notch_est), hipMemcpyDeviceToDevice, c->stream));
  } else {
    // (a run with more detect points than fit the argument segment: pinned slot → device, in stream order)
    const int hs = a->h_slot;
    a->h_slot ^= 1;
    if (a->h_ev[hs]) LSDR_HIP(hipEventSynchronize(a->h_ev[hs]));
    else LSDR_HIP(hipEventCreateWithFlags(&a->h_ev[hs], hipEventDisableTiming));
    if (a->h_cap[hs] < ndet + 1) {
      if (a->h_ifirst[hs]) (void)hipHostFree(a->h_ifirst[hs]);
      if (a->h_offsets[hs]) (void)hipHostFree(a->h_offsets[hs]);
      a->h_cap[hs] = ndet + 8;
      LSDR_HIP(hipHostMalloc((void **)&a->h_ifirst[hs], (a->h_cap[hs] + 1) * sizeof(int), hipHostMallocDefault));
      LSDR_HIP(hipHostMalloc((void **)&a->h_offsets[hs], a->h_cap[hs] * sizeof(unsigned long long), hipHostMallocDefault));
    }
    memcpy(a->h_ifirst[hs], ifirst.data(), ifirst.size() * sizeof(int));
    memcpy(a->h_offsets[hs], offs.data(), ndet * sizeof(unsigned long long));
    LSDR_HIP(hipMemcpyAsync(p_ifirst, a->h_ifirst[hs], ifirst.size() * sizeof(int), hipMemcpyHostToDevice, cs));
    LSDR_HIP(hipMemcpyAsync(p_offsets, a->h_offsets[hs], ndet * sizeof(unsigned long long), hipMemcpyHostToDevice, cs));
    LSDR_HIP(hipEventRecord(a->h_ev[hs], cs));
    LSDR_HIP(hipMemcpyAsync(a->d_scarry[nxt], a->d_scarry[cur], sizeof(notch_est), hipMemcpyDeviceToDevice, c->stream));
  }
  if (ndet) {
    int rc = cfft_dev_init(&a->fft, kN, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cfft_half, dim3((unsigned)(2 * ndet)), dim3(256), 0, cs, (const float2 *)in, (const float2 *)a->fft.d_om, p_spec,
                       (const unsigned long long *)p_offsets);
    hipLaunchKernelGGL(k_notch_peaks, dim3((unsigned)ndet), dim3(256), 0, cs, (const float2 *)p_spec, (const float2 *)a->fft.d_om,
                       (float)(1.0 / kN), ns, p_cand);
  }
  hipLaunchKernelGGL(k_notch_tables, dim3((unsigned)((ndet + 1) * ns)), dim3(256), 0, cs, (const int *)p_cand, (int)ndet, ns,
                     (const int *)(a->d_bins + cur * kMaxSlots), a->d_bins + nxt * kMaxSlots, p_reset, p_tables);
  LSDR_HIP(hipGetLastError());
  if (a->overlap) {                     // the scan of this run waits for its chain; the chain did not wait for the previous scan
    LSDR_HIP(hipEventRecord(a->ev_chain[par], cs));
    LSDR_HIP(hipStreamWaitEvent(c->stream, a->ev_chain[par], 0));
  }
  notch_scan_args sa;
  sa.in = (const float2 *)in; sa.out = (float2 *)out; sa.tables = p_tables; sa.interval_first = p_ifirst; sa.reset = p_reset;
  sa.n_intervals = (int)ifirst.size(); sa.nslots = ns; sa.n_blocks = nb;
  sa.carry = a->d_scarry[cur];       // read by every block's look-back …
  sa.totals = a->d_totals; sa.flags = a->d_flags; sa.stamp = ++a->stamp;
  if (!a->h_abort) {
    LSDR_HIP(hipHostMalloc((void **)&a->h_abort, sizeof(unsigned), hipHostMallocMapped));
    *a->h_abort = 0;
    LSDR_HIP(hipHostGetDevicePointer((void **)&a->d_abort, a->h_abort, 0));
  }
  if (*a->h_abort) {     // an earlier run's look-back gave up: its output (and everything after it) is not to be trusted
    lsdr_set_error("auto_notch(scan): a wave-block's look-back timed out in run %u (workgroups not started in index order?) — use LSDR_NOTCH_EXACT", *a->h_abort);
    return LSDR_E_UNSUPPORTED;
  }
  sa.abort_flag = a->d_abort;
  {
    const double av = 1.0 - (double)a->k;
    sa.C.k = a->k; sa.C.omk = 1 - a->k; sa.C.gain = a->gain;
    for (int j = 0; j <= kScanPer; ++j) sa.C.apow[j] = (float)pow(av, j);
    for (int m = 0; m < 7; ++m) sa.C.apow16[m] = (float)pow(av, 16.0 * (1 << m));
    sa.C.a1024 = (float)pow(av, 1024); sa.C.a2048 = (float)pow(av, 2048); sa.C.a3072 = (float)pow(av, 3072);
    sa.C.a4096 = (float)pow(av, 4096); sa.C.a8192 = (float)pow(av, 8192); sa.C.a12288 = (float)pow(av, 12288);
  }
  sa.carry_out = a->d_scarry[nxt];   // … written by the last block
  hipEvent_t *tp = nullptr;
  if (a->timing) {
    tp = a->tev[a->timed_runs % lsdr_auto_notch::kTimed];
    if (!tp[0]) { LSDR_HIP(hipEventCreate(&tp[0])); LSDR_HIP(hipEventCreate(&tp[1])); }
    LSDR_HIP(hipEventRecord(tp[0], c->stream));
  }
  switch (ns) {
    case 1: notch_scan_launch<1>(c->stream, (unsigned)nb, sa); break;
    case 2: notch_scan_launch<2>(c->stream, (unsigned)nb, sa); break;
    case 3: notch_scan_launch<3>(c->stream, (unsigned)nb, sa); break;
    case 4: notch_scan_launch<4>(c->stream, (unsigned)nb, sa); break;
    default: lsdr_set_error("auto_notch: LSDR_NOTCH_SCAN supports 1 to 4 slots"); return LSDR_E_UNSUPPORTED;
  }
  LSDR_HIP(hipGetLastError());
  if (tp) { LSDR_HIP(hipEventRecord(tp[1], c->stream)); ++a->timed_runs; }
  if (a->overlap) LSDR_HIP(hipEventRecord(a->ev_scan[par], c->stream));
  a->scarry_cur = nxt;
  return LSDR_OK;
}

// refresh the host mirrors (bins, estimators) of a scan-mode notch
static int notch_scan_pull(lsdr_auto_notch *a) {
  if (!a->scan_started) return LSDR_OK;
  if (a->side) LSDR_HIP(hipStreamSynchronize(a->side));
  LSDR_HIP(hipMemcpyAsync(a->bins, a->d_bins + a->scarry_cur * kMaxSlots, kMaxSlots * sizeof(int), hipMemcpyDeviceToHost, a->ctx->stream));
  LSDR_HIP(hipMemcpyAsync(&a->est, a->d_scarry[a->scarry_cur], sizeof(notch_est), hipMemcpyDeviceToHost, a->ctx->stream));
  LSDR_HIP(hipStreamSynchronize(a->ctx->stream));
  if (a->h_abort && *a->h_abort) {     // every point where the host has waited for the stream anyway: say so NOW, not at the next run
    lsdr_set_error("auto_notch(scan): a wave-block's look-back timed out in run %u — the output from that run on is not valid (use LSDR_NOTCH_EXACT)", *a->h_abort);
    return LSDR_E_UNSUPPORTED;
  }
  return LSDR_OK;
}

extern "C" {

int lsdr_cfft_run(lsdr_ctx *c, int n, int reverse, const lsdr_cf32 *in_dev, lsdr_cf32 *out_host) {
  LSDR_ARG(c && in_dev && out_host && n >= 2 && n <= 8192 && (n & (n - 1)) == 0);
  LSDR_HIP(hipSetDevice(c->device));
  cfft_dev f;
  int rc = cfft_dev_init(&f, n, reverse != 0);
  if (!rc) rc = cfft_dev_run(c, &f, in_dev, out_host);
  cfft_dev_free(&f);
  return rc;
}

int lsdr_cfft_host(int n, lsdr_cf32 *data, int reverse) {
  LSDR_ARG(data && n >= 1 && (n & (n - 1)) == 0);
  cfft_host(n, data, reverse != 0);
  return LSDR_OK;
}

int lsdr_auto_notch_create(lsdr_ctx *c, int nslots, float setpoint, lsdr_auto_notch **out) {
  LSDR_ARG(c && out && nslots >= 0 && nslots <= kMaxSlots);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_auto_notch *a = new lsdr_auto_notch();
  a->ctx = c; a->nslots = nslots;
  a->decimation = 1024 * 4096; a->k = (float)0.002;   // sdr.h:54
  a->phase = 0; a->gain = 1; a->agc_rms_setpoint = setpoint;
  for (int s = 0; s < kMaxSlots; ++s) a->bins[s] = -1;
  a->any_active = false;
  a->expj.assign((size_t)(nslots > 0 ? nslots : 1) * kN, lsdr_cf32{0.f, 0.f});
  memset(&a->est, 0, sizeof(a->est));
  LSDR_HIP(hipMalloc((void **)&a->d_expj, (size_t)(nslots > 0 ? nslots : 1) * kN * sizeof(float2)));
  LSDR_HIP(hipMalloc((void **)&a->d_carry, sizeof(notch_est)));
  a->d_begin = a->d_end = nullptr; a->tiles_cap = 0;
  a->last_tiles = a->last_bad = 0;
  a->mode = LSDR_NOTCH_EXACT; a->scan_started = false;
  a->d_scarry[0] = a->d_scarry[1] = nullptr; a->scarry_cur = 0; a->d_bins = nullptr;
  a->d_offsets = nullptr; a->d_spec = nullptr; a->d_cand = nullptr; a->d_ibins = nullptr; a->d_reset = nullptr; a->d_ifirst = nullptr;
  for (int i = 0; i < 2; ++i) { a->h_ifirst[i] = nullptr; a->h_offsets[i] = nullptr; a->h_cap[i] = 0; a->h_ev[i] = nullptr; }
  a->h_slot = 0;
  a->timing = false; a->timed_runs = 0; for (auto &pr : a->tev) pr[0] = pr[1] = nullptr;
  a->det_cap = 0; a->d_tables = nullptr; a->tables_cap = 0; a->d_totals = nullptr; a->d_flags = nullptr; a->blocks_cap = 0; a->stamp = 0; a->h_abort = nullptr; a->d_abort = nullptr;
  a->overlap = false; a->side = nullptr; a->ev_chain[0] = a->ev_chain[1] = a->ev_scan[0] = a->ev_scan[1] = nullptr; a->run_no = 0;
  *out = a;
  return LSDR_OK;
}
int lsdr_auto_notch_set_mode(lsdr_auto_notch *a, int mode) {
  LSDR_ARG(a && (mode == LSDR_NOTCH_EXACT || mode == LSDR_NOTCH_SCAN));
  if (mode == a->mode) return LSDR_OK;
  if (a->scan_started) { lsdr_set_error("auto_notch: the mode cannot change once the scan mode has processed data"); return LSDR_E_ARG; }
  if (mode == LSDR_NOTCH_SCAN && (a->agc_rms_setpoint != 0 || a->nslots < 1 || a->nslots > 4)) {
    lsdr_set_error("auto_notch: LSDR_NOTCH_SCAN needs 1 to 4 slots and no AGC set point (leandvb's configuration)");
    return LSDR_E_UNSUPPORTED;
  }
  a->mode = mode;
  return LSDR_OK;
}
void lsdr_auto_notch_destroy(lsdr_auto_notch *a) {
  if (!a) return;
  (void)hipStreamSynchronize(a->ctx->stream);
  (void)hipFree(a->d_scarry[0]); (void)hipFree(a->d_scarry[1]); (void)hipFree(a->d_bins); (void)hipFree(a->d_offsets); (void)hipFree(a->d_spec);
  (void)hipFree(a->d_cand); (void)hipFree(a->d_ibins); (void)hipFree(a->d_reset); (void)hipFree(a->d_ifirst); (void)hipFree(a->d_tables);
  (void)hipFree(a->d_totals); (void)hipFree(a->d_flags);
  if (a->h_abort) (void)hipHostFree(a->h_abort);
  if (a->side) {
    (void)hipStreamSynchronize(a->side);
    for (int i = 0; i < 2; ++i) { (void)hipEventDestroy(a->ev_chain[i]); (void)hipEventDestroy(a->ev_scan[i]); }
    (void)hipStreamDestroy(a->side);
  }
  for (auto &pr : a->tev) { if (pr[0]) (void)hipEventDestroy(pr[0]); if (pr[1]) (void)hipEventDestroy(pr[1]); }
  for (int i = 0; i < 2; ++i) {
    if (a->h_ifirst[i]) (void)hipHostFree(a->h_ifirst[i]);
    if (a->h_offsets[i]) (void)hipHostFree(a->h_offsets[i]);
    if (a->h_ev[i]) (void)hipEventDestroy(a->h_ev[i]);
  }
  (void)hipFree(a->d_expj); (void)hipFree(a->d_carry); (void)hipFree(a->d_begin); (void)hipFree(a->d_end);
  cfft_dev_free(&a->fft);
  delete a;
}
int lsdr_auto_notch_set(lsdr_auto_notch *a, int decimation, float k) {
  LSDR_ARG(a && decimation >= 1);
  a->decimation = decimation; a->k = k;
  return LSDR_OK;
}
int lsdr_auto_notch_slot_bin(const lsdr_auto_notch *a, int slot) {
  if (!a || slot < 0 || slot >= a->nslots) return -1;
  if (a->mode == LSDR_NOTCH_SCAN) (void)notch_scan_pull(const_cast<lsdr_auto_notch *>(a));
  return a->bins[slot];
}
#ifdef LSDR_MEASURE
// Test hook of the measure build only (tools/notch_poison_stress.py, run by tests/test_gpu_notch.py): fill the scan mode's hand-off buffers — every wave-block total and every flag —
// with garbage, as a stale or torn hand-off would leave them.  A correct protocol never reads a total whose flag does not carry
// the CURRENT run's stamp, so the next run's output must not change by a bit.
int lsdr_auto_notch_debug_poison(lsdr_auto_notch *a) {
  LSDR_ARG(a);
  if (!a->d_totals || !a->blocks_cap) return LSDR_OK;
  const size_t n = (size_t)kMaxSlots * a->blocks_cap * kWavesPerBlock;
  LSDR_HIP(hipMemsetAsync(a->d_totals, 0x7f, n * sizeof(float2), a->ctx->stream));     // 3.39e38: one of these in a carry-in is no rounding error
  LSDR_HIP(hipMemsetAsync(a->d_flags, 0xee, n * sizeof(unsigned), a->ctx->stream));     // no stamp this side of 4·10^9 runs
  return LSDR_OK;
}
#endif

// Opt-in: the detect chain of a run (it reads the run's INPUT) goes to a side stream, where it overlaps the previous run's
// k_notch_scan.  The side stream does not wait for earlier work on the context's stream, so the caller promises that an input
// buffer is complete when lsdr_auto_notch_run is called with it (a resident capture; a buffer whose producer has been waited for).
int lsdr_auto_notch_set_overlap(lsdr_auto_notch *a, int on) {
  LSDR_ARG(a);
  if (a->scan_started) {
    LSDR_HIP(hipStreamSynchronize(a->ctx->stream));
    if (a->side) LSDR_HIP(hipStreamSynchronize(a->side));
  }
  a->overlap = on != 0;
  a->run_no = 0;
  return LSDR_OK;
}

// 0 while every look-back of every queued run so far has been served; the stamp of the first run whose did not otherwise
// (meaningful after the runs have executed: synchronise the context first)
int lsdr_auto_notch_check(lsdr_auto_notch *a, unsigned *aborted_run) {
  LSDR_ARG(a && aborted_run);
  *aborted_run = a->h_abort ? *a->h_abort : 0u;
  return LSDR_OK;
}

int lsdr_auto_notch_scan_time(lsdr_auto_notch *a, int enable, float *avg_ms, unsigned *launches) {
  LSDR_ARG(a);
  if (avg_ms) *avg_ms = 0.f;
  if (launches) *launches = 0;
  if (a->timing && a->timed_runs && (avg_ms || launches)) {
    LSDR_HIP(hipStreamSynchronize(a->ctx->stream));
    const unsigned n = a->timed_runs < (unsigned)lsdr_auto_notch::kTimed ? a->timed_runs : (unsigned)lsdr_auto_notch::kTimed;
    double sum = 0;
    for (unsigned i = 0; i < n; ++i) { float ms = 0; LSDR_HIP(hipEventElapsedTime(&ms, a->tev[i][0], a->tev[i][1])); sum += ms; }
    if (avg_ms) *avg_ms = (float)(sum / n);
    if (launches) *launches = n;
  }
  a->timing = enable != 0;
  a->timed_runs = 0;
  return LSDR_OK;
}

int lsdr_auto_notch_stats(const lsdr_auto_notch *a, unsigned *tiles, unsigned *bad) {
  LSDR_ARG(a);
  if (tiles) *tiles = a->last_tiles;
  if (bad) *bad = a->last_bad;
  return LSDR_OK;
}

int lsdr_auto_notch_run(lsdr_auto_notch *a, const lsdr_cf32 *in, size_t n_in, lsdr_cf32 *out, size_t cap_out,
                        size_t *consumed, size_t *produced) {
  LSDR_ARG(a && consumed && produced);
  *consumed = 0; *produced = 0;
  size_t nb = (n_in < cap_out ? n_in : cap_out) / kN;   // while readable>=4096 && writable>=4096, sdr.h:65
  if (!nb) return LSDR_OK;
  LSDR_ARG(in && out);
  lsdr_ctx *c = a->ctx;
  LSDR_HIP(hipSetDevice(c->device));
  a->last_tiles = a->last_bad = 0;
  if (a->mode == LSDR_NOTCH_SCAN) {
    int rc = notch_run_scan(a, in, out, nb);
    if (rc) return rc;
    *consumed = nb * kN;
    *produced = nb * kN;
    return LSDR_OK;
  }
  // Split at the blocks where detect() fires (phase += 4096; if phase >= decimation …, sdr.h:66-70).
  size_t b = 0;
  while (b < nb) {
    // blocks until the next detect: detect fires at the START of block j when phase + 4096·(j−b+1) ≥ decimation
    long long until = ((long long)a->decimation - a->phase + kN - 1) / kN - 1;   // blocks processed before the detecting block
    if (until < 0) until = 0;
    size_t run = (size_t)until < nb - b ? (size_t)until : nb - b;
    int rc = notch_process(a, in + b * kN, out + b * kN, run);
    if (rc) return rc;
    a->phase += (int)(run * kN);
    b += run;
    if (b < nb) {   // this block triggers detect() on its own input, then is processed with the new slots
      a->phase += kN;
      if (a->phase >= a->decimation) {
        a->phase -= a->decimation;
        std::vector<lsdr_cf32> blk(kN);
        LSDR_HIP(hipMemcpyAsync(blk.data(), in + b * kN, kN * sizeof(lsdr_cf32), hipMemcpyDeviceToHost, c->stream));
        LSDR_HIP(hipStreamSynchronize(c->stream));
        std::vector<lsdr_cf32> spec(kN);
        rc = cfft_dev_init(&a->fft, kN, true);
        if (rc) return rc;
        rc = cfft_dev_run(c, &a->fft, in + b * kN, spec.data());
        if (rc) return rc;
        notch_detect(a, blk.data(), spec);
      }
      rc = notch_process(a, in + b * kN, out + b * kN, 1);
      if (rc) return rc;
      b += 1;
    }
  }
  LSDR_HIP(hipStreamSynchronize(c->stream));
  *consumed = nb * kN;
  *produced = nb * kN;
  return LSDR_OK;
}

// ------------------------------------------------------------------ cnr_fft
int lsdr_cnr_fft_create(lsdr_ctx *c, float bandwidth, int nfft, lsdr_cnr_fft **out) {
  LSDR_ARG(c && out && nfft >= 2 && (nfft & (nfft - 1)) == 0);
  if (bandwidth > 0.25) { lsdr_set_error("CNR estimator requires Fsampling > 4x Fsignal"); return LSDR_E_ARG; }   // sdr.h:1282-1283
  lsdr_cnr_fft *f = new lsdr_cnr_fft();
  f->ctx = c; f->bandwidth = bandwidth; f->nfft = nfft;
  f->decimation = 1048576; f->kavg = (float)0.1; f->phase = 0;
  *out = f;
  return LSDR_OK;
}
void lsdr_cnr_fft_destroy(lsdr_cnr_fft *f) { if (f) { cfft_dev_free(&f->fft); delete f; } }
int lsdr_cnr_fft_set(lsdr_cnr_fft *f, int decimation, float kavg) {
  LSDR_ARG(f && decimation >= 1);
  f->decimation = decimation; f->kavg = kavg;
  return LSDR_OK;
}

int lsdr_cnr_fft_run(lsdr_cnr_fft *f, float freq_tap, float tap_multiplier, const lsdr_cf32 *in, size_t n_in,
                     float *cnr_out, size_t cap_out, size_t *consumed, size_t *produced) {
  LSDR_ARG(f && consumed && produced);
  *consumed = 0; *produced = 0;
  const int N = f->nfft;
  size_t pos = 0, nout = 0;
  lsdr_ctx *c = f->ctx;
  // while in.readable()>=fft.n && out.writable()>=1 (sdr.h:1292)
  while (n_in - pos >= (size_t)N && nout < cap_out) {
    // skip ahead over the blocks that only advance the phase
    long long until = ((long long)f->decimation - f->phase + N - 1) / N - 1;
    if (until < 0) until = 0;
    size_t avail = (n_in - pos) / N;
    if ((size_t)until >= avail) { f->phase += (int)(avail * N); pos += avail * N; break; }
    f->phase += (int)(until * N);
    pos += (size_t)until * N;
    f->phase += N;
    if (f->phase >= f->decimation) {   // do_cnr, sdr.h:1303-1332
      f->phase -= f->decimation;
      LSDR_ARG(in && cnr_out);
      std::vector<lsdr_cf32> data(N);
      { int rc = cfft_dev_init(&f->fft, N, true); if (rc) return rc; rc = cfft_dev_run(c, &f->fft, in + pos, data.data()); if (rc) return rc; }
      const float center_freq = freq_tap * tap_multiplier;
      const int icf = (int)floor((double)(center_freq * N) + 0.5);
      std::vector<float> power(N);
      for (int i = 0; i < N; ++i) power[i] = data[i].re * data[i].re + data[i].im * data[i].im;
      if (f->avgpower.empty()) f->avgpower = power;
      for (int i = 0; i < N; ++i) f->avgpower[i] = f->avgpower[i] * (1 - f->kavg) + power[i] * f->kavg;
      const int bwslots = (int)((f->bandwidth / 4) * N);
      if (bwslots) {
        auto avgslots = [&](int i0, int i1) {
          float s = 0;
          for (int i = i0; i <= i1; ++i) s += f->avgpower[i & (N - 1)];
          return s / (i1 - i0 + 1);
        };
        const float c2plusn2 = avgslots(icf - bwslots, icf + bwslots);
        const float n2 = (avgslots(icf - bwslots * 4, icf - bwslots * 3) + avgslots(icf + bwslots * 3, icf + bwslots * 4)) / 2;
        const float c2 = c2plusn2 - n2;
        cnr_out[nout++] = (c2 > 0 && n2 > 0) ? 10 * logf(c2 / n2) / logf(10) : -50;
      }
    }
    pos += N;
  }
  *consumed = pos;
  *produced = nout;
  return LSDR_OK;
}

// ------------------------------------------------------------------ spectrum
int lsdr_spectrum_create(lsdr_ctx *c, lsdr_spectrum **out) {
  LSDR_ARG(c && out);
  lsdr_spectrum *f = new lsdr_spectrum();
  f->ctx = c; f->decimation = 1048576; f->kavg = (float)0.1; f->phase = 0;   // sdr.h:1353
  *out = f;
  return LSDR_OK;
}
void lsdr_spectrum_destroy(lsdr_spectrum *f) { if (f) { cfft_dev_free(&f->fft); delete f; } }
int lsdr_spectrum_set(lsdr_spectrum *f, int decimation, float kavg) {
  LSDR_ARG(f && decimation >= 1);
  f->decimation = decimation; f->kavg = kavg;
  return LSDR_OK;
}

int lsdr_spectrum_run(lsdr_spectrum *f, const lsdr_cf32 *in, size_t n_in, float *rows_out, size_t cap_rows,
                      size_t *consumed, size_t *produced) {
  LSDR_ARG(f && consumed && produced);
  *consumed = 0; *produced = 0;
  const int N = 1024;
  size_t pos = 0, nout = 0;
  lsdr_ctx *c = f->ctx;
  // while in.readable()>=fft.n && out.writable()>=1 (sdr.h:1362)
  while (n_in - pos >= (size_t)N && nout < cap_rows) {
    // blocks that only advance the phase are skipped arithmetically
    long long until = ((long long)f->decimation - f->phase + N - 1) / N - 1;
    if (until < 0) until = 0;
    size_t avail = (n_in - pos) / N;
    if ((size_t)until >= avail) { f->phase += (int)(avail * N); pos += avail * N; break; }
    f->phase += (int)(until * N);
    pos += (size_t)until * N;
    f->phase += N;
    if (f->phase >= f->decimation) {   // do_spectrum, sdr.h:1374-1396
      f->phase -= f->decimation;
      LSDR_ARG(in && rows_out);
      std::vector<lsdr_cf32> data(N);
      { int rc = cfft_dev_init(&f->fft, N, true); if (rc) return rc; rc = cfft_dev_run(c, &f->fft, in + pos, data.data()); if (rc) return rc; }
      std::vector<float> power(N);
      for (int i = 0; i < N; ++i) power[i] = (float)data[i].re * data[i].re + (float)data[i].im * data[i].im;
      if (f->avgpower.empty()) f->avgpower = power;
      for (int i = 0; i < N; ++i) f->avgpower[i] = f->avgpower[i] * (1 - f->kavg) + power[i] * f->kavg;
      float *row = rows_out + nout * N;
      for (int i = 0; i < N / 2; ++i) {
        row[i] = 10 * log10f(f->avgpower[N / 2 + i]);
        row[N / 2 + i] = 10 * log10f(f->avgpower[i]);
      }
      ++nout;
    }
    pos += N;
  }
  *consumed = pos;
  *produced = nout;
  return LSDR_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ notch_fir: fused auto_notch (one slot) + fir_filter
struct lsdr_notch_fir {
  lsdr_ctx *ctx;
  int N, D, decimation, phase;
  float k, scale;
  int wpc;
  unsigned long long F, A;          // stream positions: fir_filter's read pointer (= samples consumed), the notch's frontier (multiple of 4096)
  std::vector<float> coeffs; float freq; bool retap;      // fir_filter's prototype taps, its current shift (dsp.h:271-280), "re-shifted since the last run"
  float2 *d_coeffs;                 // the shifted taps, scaled
  nf_state *d_state;               // [2]: a run reads one record and leaves the next run's in the other (run parity)
  int *d_cand; float2 *d_spec;
  // per-run tables, two sets used alternately (run parity): with lsdr_notch_fir_set_overlap run k+1's detect chain and filter pass are
  // under way while run k's tail still reads its own
  unsigned *d_tile_first; int *d_ivbin; unsigned char *d_changed; float2 *d_ivP, *d_ivrho; float *d_ivtab;
  int *d_bin_carry;                 // [2] the carried bin, ping-pong between consecutive runs' k_nf_taps
  float2 *d_r[2]; size_t r_cap[2];
  unsigned run_no;
  // lsdr_notch_fir_set_overlap: detect chain + taps and the filter pass on s_det (= s_pass since round 6: one stream), the tail (head, fix-ups, scan, state) on the
  // context's stream; events hand over between them
  bool overlap; hipStream_t s_det, s_pass; hipEvent_t ev_taps[2], ev_pass[2], ev_tail[2]; bool tail_recorded[2];
  cfft_dev fft;
  // optional timing of the filter pass alone (lsdr_notch_fir_time): a ring of event pairs
  static const int kTimed = 16;
  bool timing; hipEvent_t tev[kTimed][2]; unsigned timed_runs;
};

static int nf_sync_all(lsdr_notch_fir *h);
static int nf_create_body(lsdr_notch_fir *h, const lsdr_notch_fir_cfg *cfg);
// fir_filter::set_freq (dsp.h:271-280) for the fused block: shifted taps (host libm, as the reference), the fused scaler on them
static int nf_upload_taps(lsdr_notch_fir *h) {
  std::vector<lsdr_cf32> sc(h->coeffs.size());
  lsdr::fir_shift_coeffs((unsigned)h->coeffs.size(), h->coeffs.data(), h->freq, sc.data());
  for (auto &v : sc) { v.re *= h->scale; v.im *= h->scale; }
  LSDR_TRY(nf_sync_all(h));                      // (rare: queued runs may still read the old taps)
  LSDR_HIP(hipMemcpy(h->d_coeffs, sc.data(), sc.size() * sizeof(float2), hipMemcpyHostToDevice));
  return LSDR_OK;
}

extern "C" {

int lsdr_notch_fir_create(lsdr_ctx *c, const lsdr_notch_fir_cfg *cfg, lsdr_notch_fir **out) {
  LSDR_ARG(c && cfg && out && cfg->coeffs_host && cfg->ncoeffs >= 1);
  if (cfg->nslots != 1 || cfg->decim != (unsigned)kNfD || cfg->ncoeffs + cfg->decim > (unsigned)kNfTaps) {
    lsdr_set_error("notch_fir: the fused block exists for one notch slot, decimation %d and ncoeffs ≤ %d (got %d slots, decimation %u, %u taps): "
                   "use auto_notch and fir_filter as separate blocks", kNfD, kNfTaps - kNfD, cfg->nslots, cfg->decim, cfg->ncoeffs);
    return LSDR_E_UNSUPPORTED;
  }
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_notch_fir *h = new lsdr_notch_fir();   // value-initialised: every pointer / event null until nf_create_body sets it
  h->ctx = c;
  const int rc = nf_create_body(h, cfg);
  if (rc) { lsdr_notch_fir_destroy(h); return rc; }   // every error exit frees what was allocated so far
  *out = h;
  return LSDR_OK;
}

}  // extern "C"
static int nf_create_body(lsdr_notch_fir *h, const lsdr_notch_fir_cfg *cfg) {
  h->N = (int)cfg->ncoeffs; h->D = (int)cfg->decim;
  h->decimation = cfg->notch_decimation > 0 ? cfg->notch_decimation : 1024 * 4096;   // sdr.h:56
  h->k = cfg->k > 0.f ? cfg->k : 0.002f;
  h->scale = cfg->in_scale != 0.f ? cfg->in_scale : 1.0f;
  h->phase = 0; h->F = 0; h->A = 0;
  { const char *e = getenv("LSDR_NF_WPC"); h->wpc = e && atoi(e) > 0 ? atoi(e) : 64; }   // tuning hook: workgroups per CU queued for the filter pass (oversubscribed: see k_fir_mfma_stream)
  h->d_r[0] = h->d_r[1] = nullptr; h->r_cap[0] = h->r_cap[1] = 0; h->timing = false; h->timed_runs = 0;
  h->run_no = 0; h->overlap = false; h->s_det = h->s_pass = nullptr; h->tail_recorded[0] = h->tail_recorded[1] = false;
  memset(h->ev_taps, 0, sizeof(h->ev_taps)); memset(h->ev_pass, 0, sizeof(h->ev_pass)); memset(h->ev_tail, 0, sizeof(h->ev_tail));
  memset(h->tev, 0, sizeof(h->tev));
  const size_t niv = kNfMaxDet + 1;
  h->coeffs.assign(cfg->coeffs_host, cfg->coeffs_host + cfg->ncoeffs);
  h->freq = 0.f; h->retap = false;
  LSDR_HIP(hipMalloc((void **)&h->d_coeffs, cfg->ncoeffs * sizeof(float2)));
  LSDR_TRY(nf_upload_taps(h));
  LSDR_HIP(hipMalloc((void **)&h->d_state, 2 * sizeof(nf_state)));
  nf_state s0[2]; memset(s0, 0, sizeof(s0)); s0[0].bin = s0[1].bin = -1;
  LSDR_HIP(hipMemcpy(h->d_state, s0, sizeof(s0), hipMemcpyHostToDevice));
  LSDR_HIP(hipMalloc((void **)&h->d_cand, kNfMaxDet * kMaxSlots * sizeof(int)));
  LSDR_HIP(hipMalloc((void **)&h->d_spec, (size_t)kNfMaxDet * kN * sizeof(float2)));
  LSDR_HIP(hipMalloc((void **)&h->d_tile_first, 2 * niv * sizeof(unsigned)));
  LSDR_HIP(hipMalloc((void **)&h->d_ivbin, 2 * niv * sizeof(int)));
  LSDR_HIP(hipMalloc((void **)&h->d_changed, 2 * (niv + 1)));
  LSDR_HIP(hipMalloc((void **)&h->d_ivP, 2 * niv * sizeof(float2)));
  LSDR_HIP(hipMalloc((void **)&h->d_ivrho, 2 * niv * kNfTaps * sizeof(float2)));
  LSDR_HIP(hipMalloc((void **)&h->d_ivtab, 2 * niv * kNfKs * 64 * sizeof(float)));
  LSDR_HIP(hipMalloc((void **)&h->d_bin_carry, 2 * sizeof(int)));
  { const int none[2] = {-1, -1}; LSDR_HIP(hipMemcpy(h->d_bin_carry, none, sizeof(none), hipMemcpyHostToDevice)); }
  return cfft_dev_init(&h->fft, kN, true);
}
static int nf_sync_all(lsdr_notch_fir *h) {
  if (h->s_det) LSDR_HIP(hipStreamSynchronize(h->s_det));
  if (h->s_pass) LSDR_HIP(hipStreamSynchronize(h->s_pass));
  LSDR_HIP(hipStreamSynchronize(h->ctx->stream));
  return LSDR_OK;
}
extern "C" {

// fir_filter::set_freq / the tracking of fir_filter::run (dsp.h:236-244,271-280): the taps follow the receiver's carrier estimate.  Takes
// effect with the next run (as the reference's: run() re-shifts before it filters).
int lsdr_notch_fir_set_freq(lsdr_notch_fir *h, float freq) {
  LSDR_ARG(h);
  h->freq = freq;
  h->retap = h->F != 0 || h->A != 0;             // outputs exist already: the recurrence must be re-anchored under the new taps
  return nf_upload_taps(h);
}
int lsdr_notch_fir_track(lsdr_notch_fir *h, float freq_tap, float tap_multiplier, float freq_tol, int *shifted) {
  LSDR_ARG(h);
  const float new_freq = freq_tap * tap_multiplier;          // dsp.h:237-238
  int did = 0;
  if (fabs(h->freq - new_freq) > freq_tol) { LSDR_TRY(lsdr_notch_fir_set_freq(h, new_freq)); did = 1; }
  if (shifted) *shifted = did;
  return LSDR_OK;
}
float lsdr_notch_fir_current_freq(const lsdr_notch_fir *h) { return h ? h->freq : 0.f; }

void lsdr_notch_fir_destroy(lsdr_notch_fir *h) {
  if (!h) return;
  (void)nf_sync_all(h);
  if (h->s_det) (void)hipStreamDestroy(h->s_det);
  if (h->s_pass && h->s_pass != h->s_det) (void)hipStreamDestroy(h->s_pass);
  for (int i = 0; i < 2; ++i) {
    if (h->ev_taps[i]) (void)hipEventDestroy(h->ev_taps[i]);
    if (h->ev_pass[i]) (void)hipEventDestroy(h->ev_pass[i]);
    if (h->ev_tail[i]) (void)hipEventDestroy(h->ev_tail[i]);
  }
  (void)hipFree(h->d_bin_carry); (void)hipFree(h->d_r[1]);
  (void)hipFree(h->d_coeffs); (void)hipFree(h->d_state); (void)hipFree(h->d_cand); (void)hipFree(h->d_spec);
  (void)hipFree(h->d_tile_first); (void)hipFree(h->d_ivbin); (void)hipFree(h->d_changed); (void)hipFree(h->d_ivP); (void)hipFree(h->d_ivrho);
  (void)hipFree(h->d_ivtab); (void)hipFree(h->d_r[0]);
  for (int i = 0; i < lsdr_notch_fir::kTimed; ++i) for (int j = 0; j < 2; ++j) if (h->tev[i][j]) (void)hipEventDestroy(h->tev[i][j]);
  cfft_dev_free(&h->fft);
  delete h;
}

int lsdr_notch_fir_set(lsdr_notch_fir *h, int decimation, float k) {      // auto_notch's public `decimation` and `k` (sdr.h:48-49); before the first run
  LSDR_ARG(h && decimation >= 1 && k > 0.f && k < 1.f && h->F == 0 && h->A == 0);
  h->decimation = decimation; h->k = k;
  return LSDR_OK;
}

// Opt-in: run k+1's detect chain and filter pass on streams of the block's own, next to run k's tail on the context's stream (the
// output is complete, as always, in the order of the context's stream).  The own streams do not wait for earlier work queued on the
// context: the caller promises that an input buffer is COMPLETE when lsdr_notch_fir_run is called with it (a resident capture; a buffer
// whose producer has been waited for) and stays untouched until the run has completed on the context's stream.  Same results.
int lsdr_notch_fir_set_overlap(lsdr_notch_fir *h, int on) {
  LSDR_ARG(h);
  LSDR_HIP(hipSetDevice(h->ctx->device));
  LSDR_TRY(nf_sync_all(h));
  if (on && !h->s_det) {
    // (ONE plain stream for the detect chain and the pass — the pass of run k + 1 needs its taps anyway — so that the two are back to back whatever hardware
    // queues the runtime hands out, and the tail stays on the context's stream, where it runs beside the next pass like the receiver's kernels do.  Measured
    // (bench_more.anf1, round 6): 356–358 GS/s without, 372–377 with.  Round 5 had two streams: +5 % when the runtime mapped them onto one hardware queue,
    // −8 % on queues of their own (the chain's one-workgroup kernels waiting among the pass's queued workgroups); stream priorities: −17 %.)
    LSDR_HIP(hipStreamCreateWithFlags(&h->s_det, hipStreamNonBlocking));
    h->s_pass = h->s_det;
    for (int i = 0; i < 2; ++i) {
      LSDR_HIP(hipEventCreateWithFlags(&h->ev_taps[i], hipEventDisableTiming));
      LSDR_HIP(hipEventCreateWithFlags(&h->ev_pass[i], hipEventDisableTiming));
      LSDR_HIP(hipEventCreateWithFlags(&h->ev_tail[i], hipEventDisableTiming));
    }
  }
  h->overlap = on != 0;
  h->tail_recorded[0] = h->tail_recorded[1] = false;          // (everything is idle: no stale event is waited for)
  return LSDR_OK;
}

int lsdr_notch_fir_slot_bin(lsdr_notch_fir *h) {      // auto_notch's slot bin after the runs queued so far (waits for the stream)
  if (!h) return -1;
  int bin = -1;
  if (hipMemcpyAsync(&bin, &h->d_state[h->run_no & 1u].bin, sizeof(int), hipMemcpyDeviceToHost, h->ctx->stream) != hipSuccess) return -1;
  if (hipStreamSynchronize(h->ctx->stream) != hipSuccess) return -1;
  return bin;
}

int lsdr_notch_fir_time(lsdr_notch_fir *h, int enable, float *avg_ms, unsigned *launches) {
  LSDR_ARG(h);
  LSDR_TRY(nf_sync_all(h));
  const unsigned n = h->timed_runs < (unsigned)lsdr_notch_fir::kTimed ? h->timed_runs : (unsigned)lsdr_notch_fir::kTimed;
  double sum = 0;
  for (unsigned i = 0; i < n; ++i) { float ms = 0.f; LSDR_HIP(hipEventElapsedTime(&ms, h->tev[i][0], h->tev[i][1])); sum += ms; }
  if (avg_ms) *avg_ms = n ? (float)(sum / n) : 0.f;
  if (launches) *launches = n;
  h->timing = enable != 0; h->timed_runs = 0;
  return LSDR_OK;
}

// One run = auto_notch::run over the whole 4096-sample blocks that are there (sdr.h:64-75) followed by fir_filter::run over what the notch
// has released (dsp.h:233-262): `in` is the RAW stream at fir_filter's read position; *produced = ⌊(A − F − N) / D⌋ outputs where A is the
// notch's frontier (the largest multiple of 4096 of the stream within in + n_in — fewer blocks when cap_out or the 64 detect points per run
// bind), *consumed = *produced · D.  Queued on the context's stream; nothing waits for the host.
int lsdr_notch_fir_run(lsdr_notch_fir *h, const lsdr_cf32 *in, size_t n_in, lsdr_cf32 *out, size_t cap_out, size_t *consumed, size_t *produced) {
  LSDR_ARG(h && consumed && produced);
  *consumed = 0; *produced = 0;
  lsdr_ctx *c = h->ctx;
  const unsigned long long N = (unsigned long long)h->N, D = (unsigned long long)h->D;
  {
    const double a = pow(1.0 - (double)h->k, (double)kNfLook);
    if (!(a < 1e-8)) { lsdr_set_error("notch_fir: k=%g leaves (1-k)^%d=%g of older input (use auto_notch + fir_filter)", (double)h->k, kNfLook, a); return LSDR_E_UNSUPPORTED; }
  }
  unsigned long long A = ((h->F + n_in) / kN) * kN;
  auto count_of = [&](unsigned long long a) { return a > h->F + N ? (a - h->F - N) / D : 0ull; };
  while (A > h->A && count_of(A) > cap_out) A -= kN;
  // detect points of the blocks [h->A, A): `phase += 4096; if (phase >= decimation) { phase -= decimation; detect(); }` (sdr.h:66-70)
  nf_run run;
  memset(&run, 0, sizeof(run));
  int phase = h->phase;
  {
    unsigned long long b = h->A / kN;
    for (; b < A / kN; ++b) {
      int ph = phase + kN;
      if (ph >= h->decimation) {
        if (run.ndet == kNfMaxDet) break;               // the run is cut in front of this block
        ph -= h->decimation;
        run.s_rel[run.ndet++] = b * kN - h->F;
      }
      phase = ph;
    }
    A = b * kN;
  }
  const unsigned long long count = count_of(A);
  if (A <= h->A || !count) return LSDR_OK;
  LSDR_ARG(in && out);
  LSDR_HIP(hipSetDevice(c->device));
  unsigned MW = 0;
  LSDR_TRY(lsdr_fir_stream_iv_launch(c, nullptr, 0, nullptr, 0, (unsigned)N, (unsigned)D, kNfNq, nullptr, nullptr, 0, h->wpc, &MW));
  run.mw = MW; run.count = count; run.a_prev_rel = h->A - h->F; run.a_rel = A - h->F;
  run.tile_first[0] = 0;
  for (int q = 0; q < run.ndet; ++q) {
    const unsigned long long S = run.s_rel[q];
    const unsigned long long mS = S <= N ? 0ull : (S - N + D - 1) / D;
    const unsigned long long T = (mS + MW - 1) / MW;
    unsigned long long mE = (S + D - 2) / D, mhi = T * MW ? T * MW - 1 : 0;
    if (mE > mhi) mhi = mE;
    if (mhi > count - 1) mhi = count - 1;
    run.tile_first[q + 1] = (unsigned)T;
    run.m_lo[q] = (unsigned)mS; run.m_hi[q] = (unsigned)mhi;
    if (mS <= mhi && N + mhi * D + N + 1 - S > (unsigned long long)kNfFixSpan) { lsdr_set_error("notch_fir: fix-up span"); return LSDR_E_ARG; }
  }
  const unsigned par = h->run_no & 1u;
  const size_t niv = kNfMaxDet + 1;
  if (h->r_cap[par] < count) {
    LSDR_TRY(nf_sync_all(h));
    (void)hipFree(h->d_r[par]);
    h->r_cap[par] = count + count / 8 + 1024;
    LSDR_HIP(hipMalloc((void **)&h->d_r[par], h->r_cap[par] * sizeof(float2)));
  }
  unsigned *const p_tile_first = h->d_tile_first + par * niv;
  int *const p_ivbin = h->d_ivbin + par * niv;
  unsigned char *const p_changed = h->d_changed + par * (niv + 1);
  float2 *const p_ivP = h->d_ivP + par * niv, *const p_ivrho = h->d_ivrho + par * niv * kNfTaps, *const p_r = h->d_r[par];
  float *const p_ivtab = h->d_ivtab + par * niv * kNfKs * 64;
  nf_consts C; C.k = h->k; C.omk = 1 - h->k; C.scale = h->scale; C.N = h->N;
  hipStream_t st = c->stream, sd = h->overlap ? h->s_det : st, sp = h->overlap ? h->s_pass : st;
  // detect chain + taps (sd): the tables of this parity are free once the tail of the run two runs ago is through
  if (h->overlap && h->tail_recorded[par]) LSDR_HIP(hipStreamWaitEvent(sd, h->ev_tail[par], 0));
  if (run.ndet) {
    hipLaunchKernelGGL(k_nf_cfft_half, dim3(2u * run.ndet), dim3(256), 0, sd, run, (const float2 *)in, (const float2 *)h->fft.d_om, h->d_spec);
    hipLaunchKernelGGL(k_notch_peaks, dim3((unsigned)run.ndet), dim3(256), 0, sd, (const float2 *)h->d_spec, (const float2 *)h->fft.d_om,
                       (float)(1.0 / kN), 1, h->d_cand);
  }
  hipLaunchKernelGGL(k_nf_taps, dim3((unsigned)run.ndet + 1), dim3(256), 0, sd, run, (const int *)(h->d_bin_carry + par), h->d_bin_carry + (par ^ 1u),
                     (const int *)h->d_cand, (const float2 *)h->d_coeffs, C, p_ivbin, p_changed, p_ivP, p_ivrho, p_ivtab, p_tile_first);
  LSDR_HIP(hipGetLastError());
  if (h->overlap) { LSDR_HIP(hipEventRecord(h->ev_taps[par], sd)); LSDR_HIP(hipStreamWaitEvent(sp, h->ev_taps[par], 0)); }
  // the filter pass (sp)
  hipEvent_t *tp = nullptr;
  if (h->timing) {
    tp = h->tev[h->timed_runs % lsdr_notch_fir::kTimed];
    if (!tp[0]) { LSDR_HIP(hipEventCreate(&tp[0])); LSDR_HIP(hipEventCreate(&tp[1])); }
    LSDR_HIP(hipEventRecord(tp[0], sp));
  }
  LSDR_TRY(lsdr_fir_stream_iv_launch(c, in, n_in, (lsdr_cf32 *)p_r, (size_t)count, (unsigned)N, (unsigned)D, kNfNq, p_ivtab, p_tile_first,
                                     (unsigned)run.ndet + 1, h->wpc, nullptr, sp));
  if (tp) { LSDR_HIP(hipEventRecord(tp[1], sp)); ++h->timed_runs; }
  if (h->overlap) { LSDR_HIP(hipEventRecord(h->ev_pass[par], sp)); LSDR_HIP(hipStreamWaitEvent(st, h->ev_pass[par], 0)); }
  // the tail (the context's stream): the given outputs; then r[0], the recurrence and the state for the next run in one launch
  nf_state *const st_in = h->d_state + par;
  nf_state *const st_out = h->d_state + (par ^ 1u);
  if (h->retap) { hipLaunchKernelGGL(k_nf_retap, dim3(1), dim3(256), 0, st, run, (const float2 *)in, st_in, (const float2 *)h->d_coeffs, C); h->retap = false; }
  if (run.ndet)
    hipLaunchKernelGGL(k_nf_fix, dim3((unsigned)run.ndet), dim3(256), 0, st, run, (const float2 *)in, (const nf_state *)st_in, (const int *)p_ivbin,
                       (const unsigned char *)p_changed, (const float2 *)h->d_coeffs, C, p_r);
  hipLaunchKernelGGL(k_nf_scan, dim3((unsigned)((count + kNfChunk - 1) / kNfChunk) + 1u), dim3(256), 0, st, run, (const float2 *)in, (const float2 *)p_r, (const nf_state *)st_in,
                     st_out, (const float2 *)p_ivP, (const float2 *)p_ivrho, (const int *)p_ivbin, (const unsigned char *)p_changed, C, (float2 *)out);
  LSDR_HIP(hipGetLastError());
  if (h->overlap) { LSDR_HIP(hipEventRecord(h->ev_tail[par], st)); h->tail_recorded[par] = true; }
  ++h->run_no;
  h->phase = phase; h->A = A; h->F += count * D;
  *consumed = (size_t)(count * D);
  *produced = (size_t)count;
  return LSDR_OK;
}

}  // extern "C"
