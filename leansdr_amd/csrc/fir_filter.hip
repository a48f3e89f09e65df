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
#define LSDR_STREAM_NS lsdr_fir_main
#include "fir_stream.h"
using namespace lsdr_fir;
using namespace lsdr_fir_main;

// fir_stream_sweep.hip, one translation unit per part: the stream kernel of decimation D ≡ part (mod 8), 1 … 64 without 10 and 30
// (eleven: the form with 11 tap blocks as a compile-time constant)
lsdr_fir::fir_kernel_t lsdr_fir_stream_sweep_0(unsigned D, bool cplx, bool eleven);
lsdr_fir::fir_kernel_t lsdr_fir_stream_sweep_1(unsigned D, bool cplx, bool eleven);
lsdr_fir::fir_kernel_t lsdr_fir_stream_sweep_2(unsigned D, bool cplx, bool eleven);
lsdr_fir::fir_kernel_t lsdr_fir_stream_sweep_3(unsigned D, bool cplx, bool eleven);
lsdr_fir::fir_kernel_t lsdr_fir_stream_sweep_4(unsigned D, bool cplx, bool eleven);
lsdr_fir::fir_kernel_t lsdr_fir_stream_sweep_5(unsigned D, bool cplx, bool eleven);
lsdr_fir::fir_kernel_t lsdr_fir_stream_sweep_6(unsigned D, bool cplx, bool eleven);
lsdr_fir::fir_kernel_t lsdr_fir_stream_sweep_7(unsigned D, bool cplx, bool eleven);

namespace {

#ifndef LSDR_FIR_THREADS
#define LSDR_FIR_THREADS 256
#endif
constexpr int kThreads = LSDR_FIR_THREADS;   // lanes (= outputs, R = 1) per workgroup tile


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
  // scaler: complex*T = (re*k, im*k), math.h:45-48.  `scale` is 1.0f when no scaler is
  // fused (x·1 is exact), so the staging code has no data-independent branch.
  v.x = v.x * scale;
  v.y = v.y * scale;
  return v;
}

// One tap for the R outputs of a lane.  MODE: 0 exact complex, 1 exact real-coefficient,
// 2 FMA complex, 3 FMA real.  px points at the lane's sample for output r=0.
// Coefficients are read through the CONSTANT address space: they are never written
// while a kernel runs, and this guarantees wave-uniform scalar (s_load) access even in
// the persistent kernel, where stores to `out` precede later coefficient loads.
typedef const __attribute__((address_space(4))) lsdr_v2f *cptr2;
typedef const __attribute__((address_space(4))) float *cptr1;

template <int R, int MODE>
__device__ __forceinline__ void fir_tap(cptr2 psc, cptr1 prc, const float2 *px, float (&accr)[R],
                                        float (&acci)[R]) {
  if (MODE == 0 || MODE == 2) {
    const lsdr_v2f cc = *psc;
    const float2 c = make_float2(cc.x, cc.y);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float2 x = px[r * kThreads];
      if (MODE == 0) {
        // (cr·xr − ci·xi, cr·xi + ci·xr) and the accumulation as FOUR packed operations, each product and each sum rounded on
        // its own like the reference's complex multiply-then-add (math.h:40-43, dsp.h:246-262): two v_pk_mul_f32 (the second on
        // the swapped sample — a register-half select — with the tap's imaginary part as (−ci, ci)), one v_pk_add_f32 of the two
        // product pairs, one v_pk_add_f32 into the accumulator pair.  (Written as scalar expressions the compiler spent 6.3 instructions per tap on it.)
        const lsdr_v2f xv = {x.x, x.y}, xs = {x.y, x.x};
        const lsdr_v2f p1 = (lsdr_v2f){c.x, c.x} * xv;            // (cr·xr, cr·xi)
        const lsdr_v2f p2 = (lsdr_v2f){-c.y, c.y} * xs;           // (−(ci·xi), ci·xr): the sign rides on the (scalar) tap — (−a)·b = −(a·b) exactly
        const lsdr_v2f t = p1 + p2;                               // (pr, pq)
        const lsdr_v2f acc = (lsdr_v2f){accr[r], acci[r]} + t;
        accr[r] = acc.x;
        acci[r] = acc.y;
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

// Tap phase for one staged tile: tap i ↔ u = N − i = col·D + row, visited
// col-major descending (i ascending — the reference's accumulation order).
template <int R, int MODE, int DT>
__device__ __forceinline__ void fir_taps(const fir_args &a, const float2 *lds, unsigned l, unsigned N, unsigned D,
                                         unsigned S, float (&accr)[R], float (&acci)[R]) {
#pragma unroll
  for (int r = 0; r < R; ++r) { accr[r] = 0.f; acci[r] = 0.f; }
  const float2 *base = lds + l;
  // coefficient cursors (wave-uniform → scalar loads; pointer + constant offsets
  // lets the compiler merge a column's taps into wide s_load_dwordx8/x16)
  cptr2 psc = (cptr2)a.sc;
  cptr1 prc = (cptr1)a.rc;
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
}

// ---- one tile per workgroup (any D; run-time D when DT == 0) ------------------
template <int IN_FMT, int R, int MODE, int DT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_fir(fir_args a) {
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
  // consecutive t differ by one row → conflict-free ds_write_b64.  Loads are
  // issued in batches before the LDS writes so HBM latency is paid per batch.
  const unsigned T = (mv - 1) * D + N + 1;
  const unsigned long long j0 = m0 * D;
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
  __syncthreads();

  float accr[R], acci[R];
  fir_taps<R, MODE, DT>(a, lds, l, N, D, S, accr, acci);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    unsigned lm = l + r * kThreads;
    if (lm < mv) a.out[m0 + lm] = make_float2(accr[r], acci[r]);
  }
}

// ---- persistent, software-pipelined form for compile-time D --------------------
// Each workgroup walks a strided sequence of tiles inside its XCD's contiguous
// range.  The NEXT tile's samples are requested from HBM (≈64 KB in flight per
// workgroup, held in VGPRs) before the CURRENT tile's tap phase starts and are
// written to LDS after it: HBM latency hides behind the taps within every
// workgroup, independent of what the co-resident workgroup is doing.
//
// The coefficient arrays are zero-padded on the host to a whole number of
// polyphase columns (F = D−1−N%D zeros in front, one behind; a.scp / a.rcp,
// a.ncols columns), so the tap phase is `ncols` identical straight-line blocks of
// D taps: immediate LDS offsets, wide scalar coefficient loads, no vector-memory
// instruction (nothing in the tap phase waits on the prefetch).  A zero tap adds
// ±0 to an accumulator that is never −0, i.e. changes no bit, as long as the
// sample it multiplies is finite; samples past the end of the input are replaced
// by the last valid sample for that reason.
// parts in which the next tile's loads are issued during the tap phase (1 = all before the taps)
#ifndef LSDR_FIR_SPLIT
#define LSDR_FIR_SPLIT 4
#endif

template <int IN_FMT, int R, int MODE, int DT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_fir_persist(fir_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2 *lds = reinterpret_cast<float2 *>(smem_raw);
  constexpr unsigned M = kThreads * R;
  constexpr unsigned D = DT > 0 ? DT : 1;
  constexpr unsigned S = M + kSpad;
  constexpr unsigned TMAX = (M + kSpad - 1) * D;           // staged span: all columns < S, whole rows
  constexpr int NL = (int)((TMAX + kThreads - 1) / kThreads);
  const unsigned l = threadIdx.x;
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;

  float2 v[NL];
  auto tile_of = [&](unsigned ti) { return xcd * a.tiles_per_xcd + ti; };
  auto valid = [&](unsigned ti) { return ti < a.tiles_per_xcd && tile_of(ti) < a.n_tiles; };
  // Prefetch through a per-tile buffer resource (base = first sample of the tile,
  // extent = the rest of the input): one shared 32-bit lane offset, one scalar offset
  // per load, and hardware bounds checking — reads past the end of the input return
  // 0.0, which is exactly the "finite filler" the zero-padded taps need.
  constexpr unsigned ES = IN_FMT == LSDR_IN_CU8 ? 2u : 8u;
  auto issue = [&](unsigned tile, int k_lo, int k_hi) {
    const unsigned st = tile / a.tiles_per_stream, lt = tile - st * a.tiles_per_stream;
    const unsigned long long j0 = (unsigned long long)lt * M * D;
    const unsigned long long bytes = (a.n_in - j0) * ES;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(a.ins[st])) + j0 * ES, 0,
        (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
    const unsigned voff = l * ES;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      if (k < k_lo || k >= k_hi) continue;
      if (IN_FMT == LSDR_IN_CU8) {
        unsigned short r = __builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, k * kThreads * ES, LSDR_FIR_LOAD_AUX);
        v[k] = make_float2(__uint_as_float((unsigned)r), 0.f);
      } else {
        lsdr_v2u r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, k * kThreads * ES, LSDR_FIR_LOAD_AUX);
        v[k] = make_float2(__uint_as_float(r.x), __uint_as_float(r.y));
      }
    }
  };

  unsigned ti = slot;
  if (!valid(ti)) return;
#ifdef LSDR_FIR_TRACE
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = __builtin_amdgcn_s_memtime();
#define LSDR_TR(i) { unsigned long long now__ = __builtin_amdgcn_s_memtime(); tr[i] += now__ - tc; tc = now__; }
#else
#define LSDR_TR(i)
#endif
  issue(tile_of(ti), 0, NL);
  LSDR_TR(0)
  while (true) {
    const unsigned tile = tile_of(ti);
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const unsigned t = l + k * kThreads;
      if ((k + 1) * kThreads <= TMAX || t < TMAX)
        lds[(t % D) * S + (t / D)] = finish_sample<IN_FMT>(v[k], a.in_scale);
    }
    LSDR_TR(1)
    __syncthreads();
    LSDR_TR(2)
    const unsigned tn = ti + slots;
    const bool more = valid(tn);
    // The next tile's loads are issued in LSDR_FIR_SPLIT parts, one before each group of polyphase columns: the vector-memory
    // queue takes them only as fast as HBM returns data, and a wave that is stuck issuing all 32 at once (≈ 3.3 K cycles per
    // tile) does no arithmetic meanwhile.
    float accr[R], acci[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { accr[r] = 0.f; acci[r] = 0.f; }
    cptr2 psc = (cptr2)a.scp;
    cptr1 prc = (cptr1)a.rcp;
    constexpr int P = LSDR_FIR_SPLIT;
    int col = (int)a.ncols - 1;
#pragma unroll
    for (int part = 0; part < P; ++part) {
      if (more) issue(tile_of(tn), NL * part / P, NL * (part + 1) / P);   // in flight during the tap phase
      __builtin_amdgcn_sched_barrier(0);     // keep the loads ahead of the taps
      if (part == 0) { LSDR_TR(3) }
      const int col_end = (int)a.ncols * (P - 1 - part) / P;              // this part runs columns col … col_end
      for (; col >= col_end; --col, psc += D, prc += D) {
        const float2 *px = lds + l + (D - 1) * S + (unsigned)col;
#pragma unroll
        for (int k = 0; k < (int)D; ++k) fir_tap<R, MODE>(psc + k, prc + k, px - k * (int)S, accr, acci);
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    LSDR_TR(4)
    const unsigned st = tile / a.tiles_per_stream;
    const unsigned long long m0 = (unsigned long long)(tile - st * a.tiles_per_stream) * M;
    const unsigned long long rem = a.count - m0;
    const unsigned mv = rem < M ? (unsigned)rem : M;
    float2 *const po = a.outs[st];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      unsigned lm = l + r * kThreads;
      if (lm < mv) po[m0 + lm] = make_float2(accr[r], acci[r]);
    }
    LSDR_TR(5)
    if (!more) break;
    ti = tn;
    __syncthreads();                       // LDS is rewritten next
    LSDR_TR(6)
  }
#ifdef LSDR_FIR_TRACE
  if (a.trace && (l & 63) == 0)
    for (int i = 0; i < 8; ++i) a.trace[((size_t)blockIdx.x * 4 + (l >> 6)) * 8 + i] = tr[i];
#endif
}

// ---- LSDR_FIR_MFMA: the decimating FIR on the f32 matrix pipe -----------------------
// The tap phase of k_fir_persist is bound by LDS-read + VALU issue (one ds_read_b64 and two packed operations per tap per
// lane, profiles/r03); the matrix pipe is idle chip-wide.  v_mfma_f32_16x16x4_f32 is exact f32 — D = fma(a3,b3, fma(a2,b2,
// fma(a1,b1, fma(a0,b0, C)))) with one rounding per product — so a chain of them over a K axis that walks the taps in the
// reference's order (i ascending) gives, bit for bit, what LSDR_FIR_FMA's per-lane fmaf chain gives.  The product is shaped as
// a banded Toeplitz block:
//     rows    i  = 16 consecutive outputs m = m0 + 16·G + i of "group" G,
//     columns    = 8 groups × {re, im}  (one wavefront: 128 complex outputs, ONE 16×16 accumulator tile = 4 VGPRs),
//     K slot t'  = (15 − i)·D + tap index  →  A[i][t'] = c[t' − (15−i)·D] (0 outside the taps), B[t'][G,c] = x_c[top_G − t'],
// 15·D + N slots (764 at C2: 191 MFMAs per 128 outputs; 41 % of the multiplies hit the zero band — the price of having one
// filter, not sixteen).  Zero slots are exact no-ops (fma(0, x, acc) = acc for finite x and acc ≠ −0).
// Per MFMA a lane fetches ONE dword of samples and ONE dword of coefficients from LDS (0.5 B per useful multiply-add where the
// VALU kernel reads 4), and no VALU instruction at all — the receiver's tile waves get the vector ALUs.
//
// LDS holds the tile's samples in "u-space": u = TOP − n counts samples DOWN from the newest sample of the tile (n = tile-local
// sample index), plain interleaved (re, im), 4 floats of padding after every 16·D samples so that the eight groups of a
// wavefront (group stride 16·D samples = 32·D floats ≡ 0 mod 32 banks) fall on different banks: conflict-free ds_read_b32.
// Anchoring at the top makes the walk independent of N: lane (k = l>>4, col = l&15 → g = col>>1, c = col&1) reads float
// 2·u + 4·⌊u/16D⌋ + c at u = 16·G'·D + 4·s + k (G' counts groups from the top), i.e. base + 32 B per MFMA step, +16 B once per
// D blocks of four steps.  The coefficient operand comes from a zero-padded linear table cz'[4·s + k + i·D] (LDS, written
// once per workgroup).
//
// CP = 1 (complex shifted taps, current_freq ≠ 0): two K slots per tap — (cr, x_c) then (−ci, x_{1−c}) with the sign of the
// im column's second sample flipped — the same chain as LSDR_FIR_FMA's complex kernel
//     re: fma(−ci, xi, fma(cr, xr, acc))     im: fma(ci, xr, fma(cr, xi, acc)) = fma(−ci, −xr, fma(cr, xi, acc)).
// Staging, prefetch of the next tile into registers during the MFMA phase, persistence and the XCD-aware tile walk are
// k_fir_persist's.  W = wavefronts per workgroup (tile = 128·W outputs; LDS ≈ 8·D·128·W bytes).

#ifndef LSDR_MFMA_PARTS
#define LSDR_MFMA_PARTS 8
#endif
#ifndef LSDR_MFMA_SPAN
#define LSDR_MFMA_SPAN 8
#endif
#ifndef LSDR_MFMA_ASM_STORE
#define LSDR_MFMA_ASM_STORE 1
#endif

// NLT > 0: the number of prefetch loads per lane is a compile-time constant and the staged extent is padded to exactly
// NLT·T granules — every load and every LDS write of the staging is unconditional (a tile past the end loads through an EMPTY
// buffer resource: zeros, no traffic), so the compiler can count: the LDS-write phase waits vmcnt(NLT−1−k) for load k instead of
// vmcnt(0) for all of them.  NLT = 0: run-time count (any geometry), conservative waits.
template <int DT, int W, int CP, int NLT>
__global__ __launch_bounds__(64 * W) void k_fir_mfma(fir_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr unsigned T = 64 * W;                 // lanes per workgroup
  constexpr unsigned M = 128 * W;                // outputs per tile
  constexpr unsigned D = DT;
  constexpr unsigned GW = 8 * W;                 // groups of 16 outputs per tile
  constexpr unsigned PADB = 16;                  // bytes of padding per 16·D samples
  const unsigned l = threadIdx.x;
  const unsigned NB = a.mf_blocks;               // blocks of four MFMA steps (16 K slots)
  const unsigned Uneed = 16 * (GW - 1) * D + ((16 * NB) >> CP);   // staged samples the MFMA phase reads (u-space extent), even
  const unsigned U = NLT ? 2u * NLT * T : Uneed;
  const unsigned NG = U / 2;                     // 16-byte granules (two samples)
  const unsigned NLr = NLT ? (unsigned)NLT : (NG + T - 1) / T;    // loads per lane
  constexpr int NLmax = NLT ? NLT : 32;          // register budget for the prefetch (checked on the host)
  const unsigned b_bytes = U * 8 + (U / (16 * D) + 1) * PADB;
  float *const lds_a = reinterpret_cast<float *>(smem_raw + ((b_bytes + 15) & ~15u));
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  auto tile_of = [&](unsigned ti) { return xcd * a.tiles_per_xcd + ti; };
  auto valid = [&](unsigned ti) { return ti < a.tiles_per_xcd && tile_of(ti) < a.n_tiles; };

  // coefficient operand table → LDS (constant for the launch)
  for (unsigned e = l; e < a.mf_alen; e += T) lds_a[e] = a.mf_atab[e];

  // Prefetch: granule p ↔ u ∈ {2p, 2p+1} ↔ tile-local samples n = TOP − 2p − 1 (low half of the 16 bytes), TOP − 2p.
  // Load k of lane l fetches p = (NLr−1−k)·T + (T−1−l): ascending addresses in k and in l.  The buffer resource starts at the
  // lowest sample any lane touches (clamped to the start of the stream: what lies before it is multiplied by zero taps and
  // only has to be finite — the wrapped offset is out of range and reads 0.0, like everything past the end of the input).
  const int TOP = (int)a.N + (int)(M - 1) * (int)D;
  const int nb = TOP - 2 * (int)(NLr * T) + 1;   // tile-local sample of (k = 0, l = 0); ≤ 0
  lsdr_v4u v[NLmax];
  // (the buffer resource of the next tile is set up ONCE per tile, before the MFMA phase: its scalar loads and divisions must
  // not sit between the MFMAs, where an s_waitcnt lgkmcnt(0) for a.ins[st] would also wait for every LDS read in flight)
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff;
  auto aim = [&](unsigned tile, bool live) {
    const unsigned st = live ? tile / a.tiles_per_stream : 0u, lt = tile - st * a.tiles_per_stream;
    const long long j0 = (long long)lt * M * D + nb;                 // global sample of (k = 0, l = 0), may be < 0
    const long long jb = j0 < 0 ? 0 : j0;
    const unsigned long long bytes = live ? (a.n_in - (unsigned long long)jb) * 8ull : 0ull;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(a.ins[st])) + (live ? jb * 8 : 0), 0,
        (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
    voff = l * 16u - (unsigned)((jb - j0) * 8);                      // wraps (→ out of range → 0.0) before the stream start
  };
  auto issue = [&](int k_lo, int k_hi) {
#pragma unroll
    for (int k = 0; k < NLmax; ++k) {
      if (k < k_lo || k >= k_hi || (!NLT && k >= (int)NLr)) continue;
      v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (unsigned)k * (T * 16u), 0, LSDR_FIR_LOAD_AUX);
    }
  };

  unsigned ti = slot;
  if (!valid(ti)) return;
#ifdef LSDR_FIR_TRACE
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = __builtin_amdgcn_s_memtime();
#endif
  aim(tile_of(ti), true);
  issue(0, NLmax);
  LSDR_TR(0)

  // per-lane operand cursors
  const unsigned kq = l >> 4 & 3u, col = l & 15u, g = col >> 1, c = col & 1u, w = l >> 6;
  const unsigned Gp = GW - 1 - (8 * w + g);                                        // group index from the top
  const unsigned sub = CP ? (kq & 1u) : 0u;
  const unsigned b0 = Gp * (16 * D * 8 + PADB) + ((CP ? (kq >> 1) : kq) * 8) + ((c ^ sub) * 4);   // bytes
  const unsigned a0 = (kq + col * D * (1 + CP)) * 4;                               // bytes into lds_a
  const unsigned sgn = (CP && sub && c) ? 0x80000000u : 0u;
  constexpr unsigned BSTEP = CP ? 16 : 32;       // bytes of samples per MFMA step
  constexpr unsigned BUMP = CP ? 2 * D : D;      // blocks between two paddings

  while (true) {
    const unsigned tile = tile_of(ti);
    // registers → LDS (scaler fused)
#pragma unroll
    for (int k = 0; k < NLmax; ++k) {
      if (!NLT && k >= (int)NLr) continue;
      const unsigned p = (NLr - 1 - (unsigned)k) * T + (T - 1 - l);
      if (NLT || p < NG) {
        const float s = a.in_scale;
        const lsdr_v4f x = {__uint_as_float(v[k].z) * s, __uint_as_float(v[k].w) * s, __uint_as_float(v[k].x) * s,
                            __uint_as_float(v[k].y) * s};
        *reinterpret_cast<lsdr_v4f *>(smem_raw + 16u * p + PADB * (p / (8 * D))) = x;
      }
    }
    LSDR_TR(1)
    __syncthreads();
    LSDR_TR(2)
    const unsigned tn = ti + slots;
    const bool more = valid(tn);
    aim(more ? tile_of(tn) : 0u, more);

    lsdr_v4f acc = {0.f, 0.f, 0.f, 0.f};
    const char *bp = smem_raw + b0;
    const char *ap = reinterpret_cast<const char *>(lds_a) + a0;
    unsigned bump = BUMP;
    auto ldb = [&](const char *p, int i) {
      const unsigned r = *reinterpret_cast<const unsigned *>(p + i * BSTEP);
      return __uint_as_float(CP ? (r ^ sgn) : r);
    };
    auto lda = [&](const char *p, int i) { return *reinterpret_cast<const float *>(p + i * 16); };
    // Two operand sets used alternately: the reads of block b+1 are issued BEFORE the four dependent MFMAs of block b
    // (sched_barrier pins that order), so LDS latency hides behind ≈ 160 cycles of matrix work.
    float pa[2][4], pb[2][4];
    auto fetch = [&](int set) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { pa[set][i] = lda(ap, i); pb[set][i] = ldb(bp, i); }
    };
    auto advance = [&]() {
      ap += 64;
      bp += 4 * BSTEP;
      if (--bump == 0) { bp += PADB; bump = BUMP; }
    };
    auto mac4 = [&](int set) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[set][i], pb[set][i], acc, 0, 0, 0);
    };
    fetch(0);
    constexpr int P = LSDR_MFMA_PARTS;
    const unsigned npair = NB / 2;
    const unsigned nspan = npair * LSDR_MFMA_SPAN / 8;     // the next tile's loads are all issued within the first SPAN/8 of the phase
    unsigned pair = 0;
    auto run_pairs = [&](unsigned pair_end) {
      for (; pair < pair_end; ++pair) {
        // (the last block reads one block past the slots: inside the staged / table extent, never used)
        advance(); fetch(1);
        __builtin_amdgcn_sched_barrier(0);
        mac4(0);
        __builtin_amdgcn_sched_barrier(0);
        advance(); fetch(0);
        __builtin_amdgcn_sched_barrier(0);
        mac4(1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll
    for (int part = 0; part < P; ++part) {
      if (NLT || more) issue(NLmax * part / P, NLmax * (part + 1) / P);   // in flight during the MFMA phase
      __builtin_amdgcn_sched_barrier(0);
      if (part == 0) { LSDR_TR(3) }
      run_pairs(nspan * (unsigned)(part + 1) / P);
    }
    run_pairs(npair);
    if (NB & 1u) mac4(0);
    LSDR_TR(4)

    // accumulator tile → out: register r of lane (q = l>>4, col) is row 4q + r, column col.  The stores are hidden from the
    // compiler's wait-count pass (inline asm): with a store known to be pending behind the prefetch loads it turns every
    // vmcnt(n) of the LDS-write phase into vmcnt(0) — stores may retire out of order with loads on gfx9, so the count alone does
    // not tell them apart.  Unaccounted stores only make the hardware counter LARGER than the compiler assumes, i.e. its
    // vmcnt(n) waits for at least the loads it meant (loads retire in order among themselves), and the phase ends in vmcnt(0).
    const unsigned st = tile / a.tiles_per_stream;
    const unsigned long long m0 = (unsigned long long)(tile - st * a.tiles_per_stream) * M;
    float *const po = reinterpret_cast<float *>(a.outs[st]);
    const unsigned long long mrow = m0 + 16u * (8 * w + g) + 4u * kq;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (mrow + r < a.count) {
#if LSDR_MFMA_ASM_STORE
        const float val = acc[r];
        asm volatile("global_store_dword %0, %1, off" ::"v"(po + 2 * (mrow + r) + c), "v"(val) : "memory");
#else
        po[2 * (mrow + r) + c] = acc[r];
#endif
      }
    LSDR_TR(5)
    if (!more) break;
    ti = tn;
    __syncthreads();                       // LDS is rewritten next
    LSDR_TR(6)
  }
#if LSDR_MFMA_ASM_STORE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifdef LSDR_FIR_TRACE
  if (a.trace && (l & 63) == 0 && blockIdx.x < 4096)
    for (int i = 0; i < 8; ++i) a.trace[((size_t)blockIdx.x * 4 + (l >> 6)) * 8 + i] = tr[i];
#endif
}

// ---- LSDR_FIR_MFMA_BLK: block-polyphase form — a DENSE product on the matrix pipe ----------------------------
// k_fir_mfma pays for having one filter: 59 % of its multiplies hit the zero band of the Toeplitz block, and its single
// accumulator chain runs at the 40-cycle dependent-MFMA latency (trace: the MFMA phase IS 192 × 40 cycles).  Cutting the taps
// into NQ = ⌈N/D⌉ blocks of D makes the product dense.  With x_u[u] = the tile's samples counted DOWN from its newest one,
//     Z[b][q] = Σ_{r<D} c[D·q + r] · x_u[D·b + r]          rows b = blocks of D samples, columns q = tap blocks,
//     y[b]    = Σ_{q<NQ} Z[b + q][q]                        output b (counted down from the tile's last output),
// Z = X·H is one GEMM: rows = 8 sample blocks × {re, im}, K = D (30 → 32), columns = NQ (11 → 16): 61 % useful multiplies,
// and consecutive row tiles are INDEPENDENT accumulators — two are interleaved, so the pipe issues every 32 cycles.  The
// coefficient operand is 8 VGPRs for the whole launch (no LDS table); the sample operand is one conflict-free ds_read_b32
// per MFMA (row stride 2·D + PADF floats ≡ ±4·odd mod 32 banks).  Z tiles go through a 64-row LDS ring per wavefront
// ([row][q][re,im]); after every four row tiles 32 outputs × {re, im} = 64 lanes add their NQ diagonal terms (q ascending)
// and store 256 consecutive bytes.  A wavefront owns 128 rows → 128 − (NQ−1) outputs (118 at C2): 128 MFMAs per 118 outputs
// at 32 cycles where k_fir_mfma needs 192 at 40 per 128.
// Arithmetic (stated once, oracle lo_fir_filter_blk): the reference's loop with the taps in blocks of D — each block an fmaf
// chain from zero in tap order, the block sums added in block order; a fused scaler (in_scale) multiplies the TAPS (one f32
// rounding per tap) instead of the samples.  NOT the single chain of LSDR_FIR_FMA; pinned bit for
// bit to that restatement, and under the same error bound against the reference's arithmetic.
constexpr unsigned blk_padf(unsigned D) {          // floats of padding per row of D samples: (2·D + PADF) mod 32 ∈ {4,12,20,28}
  unsigned p = 0;
  while (((2 * D + p) % 8) != 4) p += 4;
  return p;
}

// NQT > 0: the number of tap blocks is a compile-time constant — the diagonal sum's addresses become immediates and its
// loops exact (PMC of the run-time form: ≈ 900 instructions per wave tile, issued one at a time by the only wave of its
// SIMD, cost more cycles than the 128 MFMAs).
template <int DT, int W, int CP, int NLT, int NQT>
__global__ __launch_bounds__(64 * W) void k_fir_mfma_blk(fir_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr unsigned T = 64 * W, D = DT, SL = 1 + CP;
  static_assert(DT % 2 == 0, "a 16-byte granule must not straddle two rows");
  constexpr unsigned KP = (D * SL + 3) / 4 * 4, KS = KP / 4;      // K slots per row (padded), MFMA steps per row tile
  constexpr unsigned PADF = blk_padf(D), ROWF = 2 * D + PADF;     // floats per LDS row
  constexpr unsigned RW = 128;                                    // rows per wavefront (16 row tiles)
  const unsigned l = threadIdx.x;
  const unsigned NQ = NQT ? (unsigned)NQT : a.mf_blocks;          // tap blocks (≤ 16)
  constexpr int NQR = NQT ? NQT : 16;                             // diagonal terms read
  const unsigned MW = RW - (NQ - 1), M = W * MW;                  // outputs per wavefront / per tile
  const unsigned R = M + NQ - 1;                                  // rows per tile
  const unsigned Uneed = (R * D + (KP / SL - D) + 1) & ~1u;       // staged samples incl. the K padding's read-ahead, even
  const unsigned U = NLT ? 2u * NLT * T : Uneed;
  const unsigned NG = U / 2;
  const unsigned NLr = NLT ? (unsigned)NLT : (NG + T - 1) / T;
  constexpr int NLmax = NLT ? NLT : 32;
  const unsigned ROWZ = 2 * (NQ | 1u);                            // floats per ring row: [q][re,im], odd pair count
  const unsigned data_bytes = ((U + D - 1) / D + 1) * ROWF * 4;
  // Z ring of a wavefront: 64 rows + rows 0…15 once more behind them (rows 64…79), so that "row (b + q) mod 64" is plain b mod 64 + q
  char *const ring = smem_raw + ((data_bytes + 15) & ~15u) + (l >> 6) * (80 * ROWZ * 4);
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  auto tile_of = [&](unsigned ti) { return xcd * a.tiles_per_xcd + ti; };
  auto valid = [&](unsigned ti) { return ti < a.tiles_per_xcd && tile_of(ti) < a.n_tiles; };

  // coefficient operand: KS registers for the whole launch (lane (k = l>>4, q = l&15) of step s: slot 4·s + k of tap block q)
  float bco[KS];
#pragma unroll
  for (unsigned s = 0; s < KS; ++s) bco[s] = a.mf_atab[s * 64 + (l & 63u)];
  if (PADF) {   // the K padding reads the row padding: finite filler, written once (staging never touches it)
    for (unsigned b = l; b < (U + D - 1) / D + 1; b += T)
      for (unsigned j = 0; j < PADF; ++j) reinterpret_cast<float *>(smem_raw)[b * ROWF + 2 * D + j] = 0.f;
  }

  const int TOP = (int)a.N + (int)(M - 1) * (int)D;
  const int nb = TOP - 2 * (int)(NLr * T) + 1;
  lsdr_v4u v[NLmax];
  // (the buffer resource of the next tile is set up ONCE per tile, before the MFMA phase: its scalar loads and divisions must
  // not sit between the MFMAs, where an s_waitcnt lgkmcnt(0) for a.ins[st] would also wait for every LDS read in flight)
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff;
  auto aim = [&](unsigned tile, bool live) {
    const unsigned st = live ? tile / a.tiles_per_stream : 0u, lt = tile - st * a.tiles_per_stream;
    const long long j0 = (long long)lt * M * D + nb;
    const long long jb = j0 < 0 ? 0 : j0;
    const unsigned long long bytes = live ? (a.n_in - (unsigned long long)jb) * 8ull : 0ull;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(a.ins[st])) + (live ? jb * 8 : 0), 0,
        (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
    voff = l * 16u - (unsigned)((jb - j0) * 8);                      // wraps (→ out of range → 0.0) before the stream start
  };
  auto issue = [&](int k_lo, int k_hi) {
#pragma unroll
    for (int k = 0; k < NLmax; ++k) {
      if (k < k_lo || k >= k_hi || (!NLT && k >= (int)NLr)) continue;
      v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (unsigned)k * (T * 16u), 0, LSDR_FIR_LOAD_AUX);
    }
  };

  unsigned ti = slot;
  if (!valid(ti)) return;
#ifdef LSDR_FIR_TRACE
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc = __builtin_amdgcn_s_memtime();
#endif
  aim(tile_of(ti), true);
  issue(0, NLmax);
  LSDR_TR(0)

  // per-lane cursors.  Sample operand: row i = l&15 → sample block β = i>>1, component c = i&1; K slot k = (l>>4)&3.
  const unsigned kq = l >> 4 & 3u, i16 = l & 15u, beta = i16 >> 1, c = i16 & 1u, w = l >> 6;
  const unsigned sub = CP ? (kq & 1u) : 0u;
  const unsigned Bw = MW * w;                                                      // first row of this wavefront
  const unsigned a0 = ((Bw + beta) * ROWF + 2 * (CP ? (kq >> 1) : kq) + (c ^ sub)) * 4;   // bytes
  const unsigned sgn = (CP && sub && c) ? 0x80000000u : 0u;
  constexpr unsigned ASTEP = CP ? 16 : 32, ATILE = 8 * ROWF * 4;
  // accumulator tile → ring: lane (q = l&15, g4 = (l>>4)&3) holds rows 4·g4 + r = (β = 2·g4 + (r>>1), c = r&1)
  const unsigned zq = l & 15u, zrow = 2 * kq;
  // diagonal sum: lane (o = (l&63)>>1, c = l&1); batch B finalises output bm = 32·B − (NQ−1) + o from rows bm + q: byte offset of
  // its q = 0 term for even / odd B (bm mod 64 depends on B's parity only), terms `dstep` bytes apart
  const unsigned ro = (l & 63u) >> 1, rc = l & 1u;
  const unsigned dstep = ROWZ * 4 + 8;
  unsigned dbase[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) dbase[par] = ((((unsigned)(32 * par - (int)(NQ - 1) + (int)ro)) & 63u) * ROWZ + rc) * 4;

  while (true) {
    const unsigned tile = tile_of(ti);
#pragma unroll
    for (int k = 0; k < NLmax; ++k) {
      if (!NLT && k >= (int)NLr) continue;
      const unsigned p = (NLr - 1 - (unsigned)k) * T + (T - 1 - l);
      if (NLT || p < NG) {     // (no scaler here: LSDR_FIR_MFMA_BLK carries in_scale on the taps)
        const lsdr_v4u x = {v[k].z, v[k].w, v[k].x, v[k].y};
        *reinterpret_cast<lsdr_v4u *>(smem_raw + 16u * p + (PADF ? 4u * PADF * ((2 * p) / D) : 0u)) = x;
      }
    }
    LSDR_TR(1)
    __syncthreads();
    LSDR_TR(2)
    const unsigned tn = ti + slots;
    const bool more = valid(tn);
    const unsigned st = tile / a.tiles_per_stream;
    const unsigned long long m0 = (unsigned long long)(tile - st * a.tiles_per_stream) * M;
    float *const po = reinterpret_cast<float *>(a.outs[st]);
    aim(more ? tile_of(tn) : 0u, more);

    const char *ap = smem_raw + a0;
    float pa[2][2][KS];                      // [set][row tile of the pair][step]
    auto fetch1 = [&](int set, int pair, int h, unsigned s) {
      const unsigned r = *reinterpret_cast<const unsigned *>(ap + (2 * pair + h) * ATILE + s * ASTEP);
      pa[set][h][s] = __uint_as_float(CP ? (r ^ sgn) : r);
    };
    lsdr_v4f acc[2][2];                      // [pair parity][row tile of the pair]
    auto to_ring = [&](int set, int pair) {
      if (zq < NQ) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const unsigned row = (16u * pair + 8u * h + zrow) & 63u;
          const lsdr_v2f lo = {acc[set][h][0], acc[set][h][1]}, hi = {acc[set][h][2], acc[set][h][3]};
          *reinterpret_cast<lsdr_v2f *>(ring + (row * ROWZ + 2 * zq) * 4) = lo;
          *reinterpret_cast<lsdr_v2f *>(ring + ((row + 1) * ROWZ + 2 * zq) * 4) = hi;
          if (((16u * pair) & 63u) == 0) {   // rows 0…15 also as rows 64…79
            *reinterpret_cast<lsdr_v2f *>(ring + ((row + 64) * ROWZ + 2 * zq) * 4) = lo;
            *reinterpret_cast<lsdr_v2f *>(ring + ((row + 65) * ROWZ + 2 * zq) * 4) = hi;
          }
        }
      }
    };
    // diagonal sum of batch B (rows ≤ 32·B + 31 are in the ring): output bm = 32·B − (NQ−1) + o needs Z[bm + q][q], q < NQ.
    // Split in two so that neither half ever waits between two MFMAs: the reads of term q, and — a pair of row tiles later —
    // the adds (q ascending) and the store.  (Run-time NQ: all 16 terms are read, branch-free — a term q ≥ NQ is whatever lies
    // behind the row — and a select keeps it out of the sum.)
    float zv[NQR];
    auto diag_read = [&](int batch, int q) {
      zv[q] = *reinterpret_cast<const float *>(ring + dbase[batch & 1] + (unsigned)q * dstep);
    };
    float ysum = 0.f;
    auto diag_add = [&](int q) {
      const float t = ysum + zv[q];
      ysum = q == 0 ? zv[0] : (NQT || (unsigned)q < NQ ? t : ysum);
    };
    auto diag_store = [&](int batch) {
      const int bm = 32 * batch - (int)(NQ - 1) + (int)ro;
      const unsigned long long m = m0 + (M - 1 - (Bw + (unsigned)bm));
      if (bm >= 0 && m < a.count) {
#if LSDR_MFMA_ASM_STORE
        asm volatile("global_store_dword %0, %1, off" ::"v"(po + 2 * m + rc), "v"(ysum) : "memory");
#else
        po[2 * m + rc] = ysum;
#endif
      }
    };
    // Everything that is not an MFMA is cut into slices and placed BETWEEN the MFMAs of a pair of row tiles (a wavefront
    // issues in order: whatever follows the 2·KS MFMAs of a pair waits for all of them, and whatever precedes them delays them).
    // During pair P: step 0 sends pair P−1's tiles to the ring; every step fetches its share of pair P+1's sample operand;
    // even P ≥ 2 reads the diagonal terms of batch P/2 − 1 (steps ≥ 1: behind the ring writes); odd P ≥ 3 adds them up and stores.
    auto at_step = [](int i, int n, int lo, int hi) { return hi > lo ? lo + i * (hi - lo) / n : lo; };   // item i of n → a step of [lo, hi)
#pragma unroll
    for (unsigned s = 0; s < KS; ++s) { fetch1(0, 0, 0, s); fetch1(0, 0, 1, s); }
    constexpr int P8 = 8;                    // one part of the next tile's loads at the head of each pair
#pragma unroll
    for (int pair = 0; pair < 8; ++pair) {
      if (NLT || more) issue(NLmax * pair / P8, NLmax * (pair + 1) / P8);
      __builtin_amdgcn_sched_barrier(0);
      if (pair == 0) { LSDR_TR(3) }
      const int set = pair & 1;
#pragma unroll
      for (unsigned s = 0; s < KS; ++s) {
        if (s == 0) {
          acc[set][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[set][0][0], bco[0], (lsdr_v4f){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          acc[set][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[set][1][0], bco[0], (lsdr_v4f){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {     // two independent accumulators interleaved: the pipe issues every 32 cycles
          acc[set][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[set][0][s], bco[s], acc[set][0], 0, 0, 0);
          acc[set][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[set][1][s], bco[s], acc[set][1], 0, 0, 0);
        }
        if (pair < 7 && !(s & 1)) {          // two steps at a time: neighbouring dwords → ds_read2_b32
#pragma unroll
          for (int h = 0; h < 2; ++h) { fetch1(set ^ 1, pair + 1, h, s); if (s + 1 < KS) fetch1(set ^ 1, pair + 1, h, s + 1); }
        }
        if (pair >= 1 && s == 0) to_ring(set ^ 1, pair - 1);
        if (pair >= 2 && !(pair & 1)) {
#pragma unroll
          for (int q = 0; q < NQR; ++q)
            if (at_step(q, NQR, KS > 1 ? 1 : 0, KS) == (int)s) diag_read(pair / 2 - 1, q);
        }
        if (pair >= 3 && (pair & 1)) {
#pragma unroll
          for (int q = 0; q < NQR; ++q)
            if (at_step(q, NQR, 0, KS) == (int)s) diag_add(q);
          if (s == KS - 1) diag_store((pair - 3) / 2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    to_ring(1, 7);
#pragma unroll
    for (int q = 0; q < NQR; ++q) diag_read(3, q);
#pragma unroll
    for (int q = 0; q < NQR; ++q) diag_add(q);
    diag_store(3);
    LSDR_TR(4)
    if (!more) break;
    ti = tn;
    __syncthreads();                         // LDS is rewritten next
    LSDR_TR(6)
  }
#if LSDR_MFMA_ASM_STORE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifdef LSDR_FIR_TRACE
  if (a.trace && (l & 63) == 0 && blockIdx.x < 4096)
    for (int i = 0; i < 8; ++i) a.trace[((size_t)blockIdx.x * 4 + (l >> 6)) * 8 + i] = tr[i];
#endif
}


// k_fir_mfma instances: decimations with a compile-time kernel × {2, 4} wavefronts per workgroup × {real, complex} taps; the C2
// decimation also with the compile-time load counts of its geometry (nl = 32 at W = 2, 31 at W = 4).
template <int DT>
fir_kernel_t pick_mfma_d(int W, bool cplx) {
  if (W == 4) return cplx ? k_fir_mfma<DT, 4, 1, 0> : k_fir_mfma<DT, 4, 0, 0>;
  return cplx ? k_fir_mfma<DT, 2, 1, 0> : k_fir_mfma<DT, 2, 0, 0>;
}
constexpr unsigned mfma_fixed_nl(unsigned D, int W) { return D == 30 ? (W == 4 ? 31u : 32u) : 0u; }
fir_kernel_t pick_mfma(unsigned D, int W, bool cplx, unsigned nl_fixed) {
  if (nl_fixed && nl_fixed == mfma_fixed_nl(D, W)) {
    if (W == 4) return cplx ? k_fir_mfma<30, 4, 1, 31> : k_fir_mfma<30, 4, 0, 31>;
    return cplx ? k_fir_mfma<30, 2, 1, 32> : k_fir_mfma<30, 2, 0, 32>;
  }
  switch (D) {
    case 4: return pick_mfma_d<4>(W, cplx);
    case 8: return pick_mfma_d<8>(W, cplx);
    case 10: return pick_mfma_d<10>(W, cplx);
    case 16: return pick_mfma_d<16>(W, cplx);
    case 30: return pick_mfma_d<30>(W, cplx);
    default: return nullptr;
  }
}
// (real taps only: the complex-tap block product is k_fir_mfma_stream's — fir_stream.h; k_fir_mfma_blk's CP = 1 branch states the
// arithmetic of rounds 3–5 — (re, −im) pairs of one tap side by side in K — and is not instantiated any more)
template <int DT>
fir_kernel_t pick_blk_d(int W) {
  if (W == 4) return k_fir_mfma_blk<DT, 4, 0, 0, 0>;
  return k_fir_mfma_blk<DT, 2, 0, 0, 0>;
}
constexpr unsigned blk_fixed_nl(unsigned D, int W) { return D == 30 ? 29u : 0u; }
fir_kernel_t pick_blk(unsigned D, int W, bool cplx, unsigned nl_fixed, unsigned nq) {
  if (cplx) return nullptr;
  if (nl_fixed && nl_fixed == blk_fixed_nl(D, W)) {          // (nl = 29 at D = 30 implies 11 tap blocks: the C2 geometry)
    if (nq == 11) return W == 4 ? k_fir_mfma_blk<30, 4, 0, 29, 11> : k_fir_mfma_blk<30, 2, 0, 29, 11>;
    return W == 4 ? k_fir_mfma_blk<30, 4, 0, 29, 0> : k_fir_mfma_blk<30, 2, 0, 29, 0>;
  }
  switch (D) {
    case 4: return pick_blk_d<4>(W);
    case 8: return pick_blk_d<8>(W);
    case 10: return pick_blk_d<10>(W);
    case 16: return pick_blk_d<16>(W);
    case 30: return pick_blk_d<30>(W);
    default: return nullptr;
  }
}
// The stream kernel of a geometry.  want_np = pairs of row tiles per wave tile asked for (k_fir_mfma_stream NP; 0 = the geometry's
// default): the C2 geometry (decimation 30, 11 tap blocks) has compile-time forms at 8 (real taps' default), 4 (folded ring: complex
// taps' default) and 6; decimations 10 and 30 run-time tap blocks at 8; every other decimation 1 … 64 comes from the sweep
// (fir_stream_sweep.hip: 8 up to D = 34, 4 unfolded above).  A want_np the geometry has no kernel for gets the default one.  k = nullptr:
// no stream kernel (D > 64).
stream_kernel pick_stream(unsigned D, bool cplx, unsigned nq, unsigned want_np = 0) {
  const char *e = getenv("LSDR_MFMA_NQT");                 // test hook: 0 forces the run-time-NQ kernels
  if (D == 30 && nq == 11 && !(e && !atoi(e))) {
    if (want_np == 4) return {cplx ? k_fir_mfma_stream<30, 1, 11, 0, 4> : k_fir_mfma_stream<30, 0, 11, 0, 4>, 4u, true};
    if (want_np == 6 && cplx) return {k_fir_mfma_stream<30, 1, 11, 0, 6>, 6u, false};
    return {cplx ? k_fir_mfma_stream<30, 1, 11> : k_fir_mfma_stream<30, 0, 11>, 8u, false};
  }
  if (D == 10 && nq == 11 && !(e && !atoi(e))) return {cplx ? k_fir_mfma_stream<10, 1, 11> : k_fir_mfma_stream<10, 0, 11>, 8u, false};
  if (D == 10) return {cplx ? k_fir_mfma_stream<10, 1, 0> : k_fir_mfma_stream<10, 0, 0>, 8u, false};
  if (D == 30) return {cplx ? k_fir_mfma_stream<30, 1, 0> : k_fir_mfma_stream<30, 0, 0>, 8u, false};
  if (D < 1 || D > kStreamMaxD) return {nullptr, 0u, false};
  static fir_kernel_t (*const part[8])(unsigned, bool, bool) = {lsdr_fir_stream_sweep_0, lsdr_fir_stream_sweep_1, lsdr_fir_stream_sweep_2, lsdr_fir_stream_sweep_3,
                                                          lsdr_fir_stream_sweep_4, lsdr_fir_stream_sweep_5, lsdr_fir_stream_sweep_6, lsdr_fir_stream_sweep_7};
  return {part[D % 8](D, cplx, nq == 11 && !(e && !atoi(e))), stream_sweep_np(D, cplx), false};
}
// geometry of a k_fir_mfma_blk launch (nb = tap blocks NQ, alen = coefficient operand floats, M = outputs per tile)
struct blk_geom { unsigned nq, alen, U, lds, nl, nl_fixed, M, ks; };
blk_geom blk_geometry(unsigned N, unsigned D, int W, bool cplx) {
  blk_geom g;
  // complex taps: the stream kernel's form only — K slots are the samples of a row as for real taps, three coefficient operands
  // (re, −im, +im parts: fir_stream.h); the register-staged kernel serves real taps
  const unsigned sl = 1, kp = (D + 3) / 4 * 4;
  g.ks = kp / 4;
  g.nq = (N + D - 1) / D;
  g.alen = (cplx ? 3 : 1) * g.ks * 64;
  if (g.nq > 16 || g.nq < 1) { g.M = 0; g.U = g.lds = g.nl = g.nl_fixed = 0; return g; }
  if (cplx || pick_blk(D, W, cplx, 0, 0) == nullptr) {      // no register-staged kernel for this decimation: tap blocks and the coefficient operand only (the stream kernel's)
    g.M = W * (128 - (g.nq - 1)); g.U = g.nl = g.nl_fixed = 0; g.lds = ~0u;
    return g;
  }
  const unsigned MW = 128 - (g.nq - 1);
  g.M = W * MW;
  const unsigned R = g.M + g.nq - 1;
  g.U = (R * D + (kp / sl - D) + 1) & ~1u;
  g.nl = (g.U / 2 + 64 * W - 1) / (64 * W);
  const char *e = getenv("LSDR_MFMA_NLT");
  g.nl_fixed = (g.nl == blk_fixed_nl(D, W) && !(e && !atoi(e))) ? g.nl : 0;
  if (g.nl_fixed) g.U = 2 * g.nl * 64 * W;
  const unsigned rowf = 2 * D + blk_padf(D);
  const unsigned data_bytes = ((g.U + D - 1) / D + 1) * rowf * 4;
  g.lds = ((data_bytes + 15) & ~15u) + W * 80 * 2 * (g.nq | 1u) * 4 + 128;   // + the branch-free diagonal reads' overrun
  return g;
}

// geometry of a k_fir_mfma launch: K slots, blocks, staged extent, LDS bytes, prefetch loads per lane
struct mfma_geom { unsigned nb, alen, U, lds, nl, nl_fixed; };
mfma_geom mfma_geometry(unsigned N, unsigned D, int W, bool cplx) {
  mfma_geom g;
  const unsigned cp = cplx ? 1 : 0, slots = (15 * D + N) << cp;
  g.nb = (slots + 15) / 16;
  g.alen = 16 * (g.nb + 1) + ((15 * D) << cp) + 4;     // one block of read-ahead past the last slot
  g.U = 16 * (8 * W - 1) * D + ((16 * g.nb) >> cp);
  g.nl = (g.U / 2 + 64 * W - 1) / (64 * W);
  const char *e = getenv("LSDR_MFMA_NLT");             // test hook: 0 forces the run-time-count kernel
  g.nl_fixed = (g.nl == mfma_fixed_nl(D, W) && !(e && !atoi(e))) ? g.nl : 0;
  if (g.nl_fixed) g.U = 2 * g.nl * 64 * W;             // staged extent padded to whole loads (k_fir_mfma NLT)
  const unsigned b_bytes = g.U * 8 + (g.U / (16 * D) + 1) * 16;
  g.lds = ((b_bytes + 15) & ~15u) + g.alen * 4 + 256;  // + the read-ahead of the top group's last block
  return g;
}

template <int IN_FMT, int R, int DT>
fir_kernel_t pick_mode(int mode, bool persist = false) {
  if (persist && DT > 0) {
    switch (mode) {
      case 0: return k_fir_persist<IN_FMT, R, 0, DT>;
      case 1: return k_fir_persist<IN_FMT, R, 1, DT>;
      case 2: return k_fir_persist<IN_FMT, R, 2, DT>;
      default: return k_fir_persist<IN_FMT, R, 3, DT>;
    }
  }
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
fir_kernel_t pick_spec(unsigned D, int mode, int *R_out, bool persist) {
#define LSDR_FIR_SPEC(DD) case DD: *R_out = spec_r(DD); return pick_mode<IN_FMT, spec_r(DD), DD>(mode, persist);
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

// The fused auto_notch + fir_filter block of notch.hip (lsdr_notch_fir_*): its matrix-pipe pass is k_fir_mfma_stream with complex taps
// that change along the stream.  `align_n` = the fir_filter's ncoeffs (output m is aligned at sample align_n + m·D of `in`, dsp.h:246-262),
// nq tap blocks of D = 30 are applied (taps align_n … nq·D − 1 reach back past the aligned window).
int lsdr_fir_stream_iv_launch(lsdr_ctx *c, const void *in, size_t n_in, lsdr_cf32 *out, size_t count, unsigned align_n, unsigned D, unsigned nq,
                              const float *iv_tabs, const unsigned *iv_tile_first, unsigned n_iv, int wpc, unsigned *outputs_per_tile, hipStream_t stream) {
  if (D != 30 || nq != 12) { lsdr_set_error("notch_fir: the fused matrix-pipe pass exists for decimation 30 with 12 tap blocks"); return LSDR_E_UNSUPPORTED; }
  static const unsigned np = getenv("LSDR_NF_NP") && atoi(getenv("LSDR_NF_NP")) == 8 ? 8u : getenv("LSDR_NF_NP") && atoi(getenv("LSDR_NF_NP")) == 6 ? 6u : 4u;   // tuning hook: rows per wave tile / 16
  const unsigned M = 16u * np - (nq - 1);
  if (outputs_per_tile) *outputs_per_tile = M;
  if (!count) return LSDR_OK;
  fir_args a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.out = (float2 *)out;
  a.n_streams = 1;
  a.ins[0] = in; a.outs[0] = (float2 *)out;
  a.N = align_n; a.D = D;
  a.count = count; a.n_in = n_in;
  const size_t n_tiles = (count + M - 1) / M;
  LSDR_ARG(n_tiles < (1ull << 31));
  a.tiles_per_stream = (unsigned)n_tiles;
  a.n_tiles = (unsigned)n_tiles;
  a.tiles_per_xcd = (unsigned)((n_tiles + 7) / 8);
  a.in_scale = 1.0f;
  a.mf_atab = iv_tabs; a.mf_alen = 3 * 8 * 64; a.mf_blocks = nq;
  a.iv_tile_first = iv_tile_first; a.n_iv = n_iv;
  { static const bool strided = getenv("LSDR_MFMA_CHUNK") && !atoi(getenv("LSDR_MFMA_CHUNK")); a.chunked = strided ? 0u : 1u; }
  fir_kernel_t k = np == 4 ? k_fir_mfma_stream<30, 1, 12, 1, 4> : np == 6 ? k_fir_mfma_stream<30, 1, 12, 1, 6> : k_fir_mfma_stream<30, 1, 12, 1>;
  const size_t lds_bytes = stream_lds(D, nq, true, np, true, true);
  unsigned grid = a.tiles_per_xcd * 8;
  const unsigned pg = (unsigned)(c->num_cu * (wpc > 0 ? wpc : 64) + 7) / 8 * 8;
  if (grid > pg) grid = pg;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds_bytes, stream ? stream : c->stream, a);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

#ifdef LSDR_FIR_TRACE
static unsigned long long *g_fir_trace = nullptr;
extern "C" int lsdr_fir_trace_read(unsigned long long *host, size_t n) {
  if (!g_fir_trace) return -1;
  return hipMemcpy(host, g_fir_trace, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

struct lsdr_fir_filter {
  lsdr_ctx *ctx;
  lsdr_fir_filter_cfg cfg;
  std::vector<float> coeffs;        // prototype (host)
  std::vector<lsdr_cf32> shifted;   // host copy of shifted_coeffs
  float2 *d_sc;                     // device: shifted coefficients
  float *d_rc;                      // device: real parts (valid when all imag == 0)
  float2 *d_scp;                    // device: column-padded copies for the persistent kernel
  float *d_rcp;
  unsigned ncols;
  bool all_real;
  float current_freq;
  int R;                            // outputs per lane
  unsigned S;                       // LDS row stride
  size_t lds_bytes;
  int force_complex;                // test hook (env LSDR_FIR_FORCE_COMPLEX)
  bool spec;                        // compile-time-D kernel in use
  bool persist;                     // persistent software-pipelined kernel (spec only)
  unsigned persist_grid;            // workgroups launched by the persistent kernel
  // LSDR_FIR_MFMA (k_fir_mfma): wavefronts per workgroup, workgroups per CU, coefficient operand tables for the real-tap and
  // the complex-tap form (d_atab[cplx], geometry mf[cplx]); mfma_ok[cplx] = that form exists for this N / D
  int mf_W, mf_wpc;
  float *d_atab[2];
  mfma_geom mf[2];
  bool mfma_ok[2];
  // LSDR_FIR_MFMA_BLK (k_fir_mfma_blk): coefficient operand tables and geometry, real / complex taps
  float *d_btab[2];
  blk_geom bk[2];
  bool blk_ok[2];
  bool stream;                      // k_fir_mfma_stream (one wavefront per workgroup, LDS-direct refill) instead of k_fir_mfma_blk
  int stream_wpc;                   // its workgroups (= wavefronts) per CU in the persistent grid
  int stream_wpc_carry;             // the same for the tiles that carry their ring rows (fir_stream.h CARRY): longer lists, one warm-up pair each
  unsigned stream_xrot;             // tiles by which XCD x's walk through its range is rotated (× x): tuning hook LSDR_MFMA_XROT, 0 = off
  unsigned stream_chunked;          // a workgroup's tiles consecutive instead of strided (LSDR_MFMA_CHUNK, read per create)
  unsigned stream_np[2];            // pairs of row tiles per wave tile, real / complex taps (k_fir_mfma_stream NP; LSDR_MFMA_NP / LSDR_MFMA_NP_CP, read per create)
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
  // column-padded copies: F = D-1-N%D zero taps in front, one behind → ncols·D taps
  const unsigned D = f->cfg.decim, F = D - 1 - N % D;
  std::vector<lsdr_cf32> scp((size_t)f->ncols * D, lsdr_cf32{0.f, 0.f});
  std::vector<float> rcp((size_t)f->ncols * D, 0.f);
  for (unsigned i = 0; i < N; ++i) { scp[F + i] = f->shifted[i]; rcp[F + i] = rc[i]; }
  LSDR_HIP(hipMemcpyAsync(f->d_scp, scp.data(), scp.size() * sizeof(float2), hipMemcpyHostToDevice, c->stream));
  LSDR_HIP(hipMemcpyAsync(f->d_rcp, rcp.data(), rcp.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  // k_fir_mfma's A operand: cz'[e'] = tap (e' − 15·D) of the real kernel; (cr, −ci) pairs of tap (e'/2 − 15·D) of the complex one
  for (int cp = 0; cp < 2; ++cp) {
    if (!f->mfma_ok[cp]) continue;
    std::vector<float> at(f->mf[cp].alen, 0.f);
    for (unsigned i = 0; i < N; ++i) {
      if (cp == 0) at[15 * D + i] = rc[i];
      else { at[2 * (15 * D + i)] = f->shifted[i].re; at[2 * (15 * D + i) + 1] = -f->shifted[i].im; }
    }
    LSDR_HIP(hipMemcpyAsync(f->d_atab[cp], at.data(), at.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    LSDR_HIP(hipStreamSynchronize(c->stream));   // `at` is pageable and dies here
  }
  // The block product's coefficient operand: lane (k = l>>4, q = l&15) of step s holds tap D·q + 4·s + k of tap block q — real taps: one
  // table of ks steps; complex taps: three (re parts, −im parts, +im parts: k_fir_mfma_stream's four MFMAs per step); zero outside the
  // block / the filter
  const float scale = f->cfg.in_scale != 0.f ? f->cfg.in_scale : 1.0f;
  for (int cp = 0; cp < 2; ++cp) {
    if (!f->blk_ok[cp]) continue;
    std::vector<float> bt(f->bk[cp].alen, 0.f);
    const unsigned ks = f->bk[cp].ks;
    for (unsigned s = 0; s < ks; ++s)
      for (unsigned ln = 0; ln < 64; ++ln) {
        const unsigned r = 4 * s + (ln >> 4), q = ln & 15;
        if (q >= f->bk[cp].nq || r >= D || D * q + r >= N) continue;
        // the fused scaler rides on the taps: one f32 rounding per tap (component)
        if (cp == 0) bt[s * 64 + ln] = rc[D * q + r] * scale;
        else {
          const float re = f->shifted[D * q + r].re * scale, im = f->shifted[D * q + r].im * scale;
          bt[s * 64 + ln] = re; bt[(ks + s) * 64 + ln] = -im; bt[(2 * ks + s) * 64 + ln] = im;
        }
      }
    LSDR_HIP(hipMemcpyAsync(f->d_btab[cp], bt.data(), bt.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    LSDR_HIP(hipStreamSynchronize(c->stream));
  }
  LSDR_HIP(hipStreamSynchronize(c->stream));
  return LSDR_OK;
}

extern "C" {

int lsdr_fir_filter_create(lsdr_ctx *c, const lsdr_fir_filter_cfg *cfg, lsdr_fir_filter **out) {
  LSDR_ARG(c && cfg && out);
  LSDR_ARG(cfg->ncoeffs >= 1 && cfg->coeffs_host && cfg->decim >= 1);
  LSDR_ARG(cfg->in_format == LSDR_IN_CF32 || cfg->in_format == LSDR_IN_CU8);
  LSDR_ARG(cfg->arith == LSDR_FIR_EXACT || cfg->arith == LSDR_FIR_FMA || cfg->arith == LSDR_FIR_MFMA || cfg->arith == LSDR_FIR_MFMA_BLK);
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
  f->persist = false;
  if (!(fg && atoi(fg)) && N / D + 2 <= kSpad && pick_spec<LSDR_IN_CF32>(D, 0, &Rs, false) != nullptr) {
    f->spec = true;
    {
      const char *fp = getenv("LSDR_FIR_PERSIST");   // test/bench hook: 0 disables, N>1 = workgroups per CU
      int wpc = fp ? atoi(fp) : 2;
      f->persist = wpc > 0;
      f->persist_grid = (unsigned)(c->num_cu * (wpc > 0 ? wpc : 2) + 7) / 8 * 8;
    }
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
  f->ncols = N / D + 1;
  LSDR_HIP(hipMalloc((void **)&f->d_scp, (size_t)f->ncols * D * sizeof(float2)));
  LSDR_HIP(hipMalloc((void **)&f->d_rcp, (size_t)f->ncols * D * sizeof(float)));
  // LSDR_FIR_MFMA: available for cf32 input at the decimations k_fir_mfma is instantiated for, while a tile (and its
  // prefetch: ≤ 32 sixteen-byte loads per lane) fits; everything else runs LSDR_FIR_FMA's kernels — the same bits.
  f->d_atab[0] = f->d_atab[1] = nullptr;
  f->mfma_ok[0] = f->mfma_ok[1] = false;
  {
    const char *ew = getenv("LSDR_MFMA_W"), *ep = getenv("LSDR_MFMA_WPC");
    f->mf_W = ew && atoi(ew) == 4 ? 4 : 2;
    f->mf_wpc = ep && atoi(ep) > 0 ? atoi(ep) : (f->mf_W == 4 || cfg->arith == LSDR_FIR_MFMA_BLK ? 2 : 3);   // more workgroups than fit at once: the resident ones fall out of step (measured: W = 2: 2 → 0.154 ms, 3 → 0.119 ms per 64 Mi)
  }
  f->d_btab[0] = f->d_btab[1] = nullptr;
  f->blk_ok[0] = f->blk_ok[1] = false;
  if (cfg->arith == LSDR_FIR_MFMA_BLK) {
    // available for cf32 input, decimations 1 … 64 (k_fir_mfma_stream; LSDR_MFMA_STREAM=0: the register-staged k_fir_mfma_blk — 4, 8, 10,
    // 16, 30), N ≤ 16·D; anything else is refused at create time (the blocked sum is its own arithmetic: there is no other kernel
    // with the same bits to fall back to)
    {
      const char *es = getenv("LSDR_MFMA_STREAM"), *ew = getenv("LSDR_MFMA_SWPC");
      f->stream = pick_stream(D, false, 0).k != nullptr && !(es && !atoi(es));
      if (!(cfg->in_format == LSDR_IN_CF32 && (f->stream || pick_blk(D, f->mf_W, false, 0, 0) != nullptr))) {
        lsdr_fir_filter_destroy(f);
        lsdr_set_error("lsdr_fir_filter_create: LSDR_FIR_MFMA_BLK has no kernel for %u taps / decimation %u / input format %d", N, D, cfg->in_format);
        return LSDR_E_ARG;
      }
      // OVERSUBSCRIBED: 96 workgroups per CU queued, each with a short tile list (3 tiles at the C2 batch), instead of a grid of exactly
      // the resident workgroups (3 per CU) that own a 99-tile list each: the dispatcher deals the work, the wavefronts of a CU fall out
      // of step, a workgroup that could not start next to the receiver's tiles costs 3 tiles, not a second round.  Same box, buffer
      // placement chosen (bench.py), strided lists, three processes each: 3 → 602–612 GS/s, 48 → 627–635, 96 → 605–624, 192 → 585–601; 6
      // (two exact rounds) → 465.  Complex taps: 4 → 402, 48 → 409.
      // CHUNKED lists (stream_chunked: a workgroup's tiles are consecutive — it stays inside one or two 2 MiB pages and re-reads its own
      // halo — instead of `slots` tiles apart): alone 5.88 → 6.02 TB/s at 48 per CU, 6.11 at 96 (0.76 of the HBM peak); C2 pipeline, two
      // processes each: strided 48: 609–644 GS/s; chunked 24 / 48 / 96 / 192: 606–634 / 619–667 / 644–651 / 602–610.
      f->stream_wpc = ew && atoi(ew) > 0 ? atoi(ew) : 96;
      // CARRY tiles (complex taps at the C2 geometry): a list starts with a warm-up pair, so longer lists — c2_offset, workgroups per CU queued
      // 32 / 48 / 64 / 96 / 128 / 192: 584 / 587 / 582 / 575.5 / 571 / 544 GS/s
      f->stream_wpc_carry = ew && atoi(ew) > 0 ? atoi(ew) : 48;
      { const char *ex = getenv("LSDR_MFMA_XROT"); f->stream_xrot = ex ? (unsigned)strtoul(ex, nullptr, 0) : 0u; }      // (read per create: A/B in one process)
      { const char *ec = getenv("LSDR_MFMA_CHUNK"); f->stream_chunked = ec ? (unsigned)atoi(ec) : 1u; }
      { const char *e0 = getenv("LSDR_MFMA_NP"), *e1 = getenv("LSDR_MFMA_NP_CP"); f->stream_np[0] = e0 && atoi(e0) == 4 ? 4u : 8u; f->stream_np[1] = e1 && (atoi(e1) == 4 || atoi(e1) == 6 || atoi(e1) == 8) ? (unsigned)atoi(e1) : 4u; }   // (asked for; pick_stream gives the geometry's default where that form does not exist)
    }
    for (int cp = 0; cp < 2; ++cp) {
      f->bk[cp] = blk_geometry(N, D, f->mf_W, cp != 0);
      // (blk_ok: the coefficient operand exists — with the stream kernel that is all; the register-staged one also needs its tile to fit)
      f->blk_ok[cp] = f->bk[cp].M > 0 && (f->stream || cp == 1 || (f->bk[cp].nl <= 32 && f->bk[cp].lds <= (size_t)160 * 1024 / (f->mf_W == 4 ? 1 : 2)));
      if (!f->blk_ok[cp]) {
        lsdr_fir_filter_destroy(f);
        lsdr_set_error("lsdr_fir_filter_create: LSDR_FIR_MFMA_BLK has no kernel for %u taps / decimation %u / input format %d", N, D, cfg->in_format);
        return LSDR_E_ARG;
      }
      LSDR_HIP(hipMalloc((void **)&f->d_btab[cp], f->bk[cp].alen * sizeof(float)));
    }
  }
  if (cfg->arith == LSDR_FIR_MFMA && cfg->in_format == LSDR_IN_CF32 && pick_mfma(D, f->mf_W, false, 0) != nullptr) {
    for (int cp = 0; cp < 2; ++cp) {
      f->mf[cp] = mfma_geometry(N, D, f->mf_W, cp != 0);
      f->mfma_ok[cp] = f->mf[cp].nl <= 32 && f->mf[cp].lds <= (size_t)160 * 1024 / (f->mf_W == 4 ? 1 : 2);
      if (f->mfma_ok[cp]) LSDR_HIP(hipMalloc((void **)&f->d_atab[cp], f->mf[cp].alen * sizeof(float)));
    }
  }
  *out = f;
  return lsdr_fir_filter_set_freq(f, 0.0f);  // fir_filter ctor ends with set_freq(0), dsp.h:230
}

void lsdr_fir_filter_destroy(lsdr_fir_filter *f) {
  if (!f) return;
  (void)hipStreamSynchronize(f->ctx->stream);
  (void)hipFree(f->d_sc);
  (void)hipFree(f->d_rc);
  (void)hipFree(f->d_scp);
  (void)hipFree(f->d_rcp);
  (void)hipFree(f->d_atab[0]);
  (void)hipFree(f->d_atab[1]);
  (void)hipFree(f->d_btab[0]);
  (void)hipFree(f->d_btab[1]);
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

static int fir_run_streams(lsdr_fir_filter *f, unsigned n_streams, const void *const *ins, size_t n_in, lsdr_cf32 *const *outs,
                           size_t cap_out, size_t *consumed, size_t *produced) {
  LSDR_ARG(f && consumed && produced && n_streams >= 1 && n_streams <= 8);
  *consumed = 0;
  *produced = 0;
  const unsigned N = f->cfg.ncoeffs, D = f->cfg.decim;
  if (n_in < N) return LSDR_OK;  // dsp.h:234
  size_t count = (n_in - N) / D;
  if (count > cap_out) count = cap_out;
  if (!count) return LSDR_OK;
  LSDR_ARG(ins && outs);
  for (unsigned i = 0; i < n_streams; ++i) LSDR_ARG(ins[i] && outs[i]);
  const bool real_taps = f->all_real && !f->force_complex;
  const bool mfma = f->cfg.arith == LSDR_FIR_MFMA && f->mfma_ok[real_taps ? 0 : 1];
  const bool blk = f->cfg.arith == LSDR_FIR_MFMA_BLK;
  if (n_streams > 1 && !mfma && !blk && !(f->spec && f->persist)) {   // only the persistent kernels take several buffers per launch
    for (unsigned i = 0; i < n_streams; ++i) {
      int rc = fir_run_streams(f, 1, ins + i, n_in, outs + i, cap_out, consumed, produced);
      if (rc) return rc;
    }
    return LSDR_OK;
  }
  const void *in = ins[0];
  lsdr_cf32 *out = outs[0];

  fir_args a;
  a.in = in;
  a.out = (float2 *)out;
  a.n_streams = n_streams;
  for (unsigned i = 0; i < 8; ++i) { a.ins[i] = i < n_streams ? ins[i] : nullptr; a.outs[i] = i < n_streams ? (float2 *)outs[i] : nullptr; }
  a.sc = f->d_sc;
  a.rc = f->d_rc;
  a.scp = f->d_scp;
  a.rcp = f->d_rcp;
  a.ncols = f->ncols;
  a.N = N; a.D = D; a.S = f->S;
  a.count = count;
  a.n_in = n_in;
  const bool stream = blk && (f->stream || !real_taps);      // (complex taps: the stream kernel, whatever LSDR_MFMA_STREAM says)
  // rows per wave tile / 16 of the stream kernel: the requested one if that kernel exists for this geometry
  const stream_kernel sk = stream ? pick_stream(D, !real_taps, f->bk[real_taps ? 0 : 1].nq, f->stream_np[real_taps ? 0 : 1]) : stream_kernel{nullptr, 8u, false};
  const unsigned snp = sk.np;
  // the stream kernel's tiles: 16·np rows; with CARRY (fir_stream.h) every row of a tile is an output and the first tile of a stream starts
  // nq − 1 rows in front of output 0, without it a tile re-reads the nq − 1 rows in front of its outputs
  const unsigned snq = f->bk[real_taps ? 0 : 1].nq;
  const bool carry = stream && sk.carry();
  const unsigned M = stream ? (carry ? 16u * snp : 16u * snp - (snq - 1)) : blk ? f->bk[real_taps ? 0 : 1].M : mfma ? 128u * f->mf_W : kThreads * f->R;
  size_t n_tiles = (count + (carry ? snq - 1 : 0) + M - 1) / M;
  LSDR_ARG(n_tiles * n_streams < (1ull << 31));
  a.tiles_per_stream = (unsigned)n_tiles;
  n_tiles *= n_streams;
  a.n_tiles = (unsigned)n_tiles;
  a.tiles_per_xcd = (unsigned)((n_tiles + 7) / 8);
  a.in_scale = f->cfg.in_scale != 0.f ? f->cfg.in_scale : 1.0f;
  a.mf_atab = nullptr; a.mf_alen = 0; a.mf_blocks = 0;
  a.iv_tile_first = nullptr; a.n_iv = 0; a.xcd_rot = f->stream_xrot; a.chunked = f->stream_chunked;
  a.trace = nullptr;
#ifdef LSDR_FIR_TRACE
  {
    static unsigned long long *d_trace = nullptr;
    if (!d_trace) LSDR_HIP(hipMalloc((void **)&d_trace, 4096 * 4 * 8 * sizeof(unsigned long long)));
    LSDR_HIP(hipMemsetAsync(d_trace, 0, 4096 * 4 * 8 * sizeof(unsigned long long), f->ctx->stream));
    a.trace = d_trace;
    g_fir_trace = d_trace;
  }
#endif

  if (mfma || blk) {
    const int cp = real_taps ? 0 : 1;
    a.mf_atab = blk ? f->d_btab[cp] : f->d_atab[cp];
    a.mf_alen = blk ? f->bk[cp].alen : f->mf[cp].alen;
    a.mf_blocks = blk ? f->bk[cp].nq : f->mf[cp].nb;
    static const char *const enq = getenv("LSDR_MFMA_NQT");
    const unsigned nqk = blk && !(enq && !atoi(enq)) ? f->bk[cp].nq : 0;
    fir_kernel_t k = stream ? sk.k : blk ? pick_blk(D, f->mf_W, cp != 0, f->bk[cp].nl_fixed, nqk) : pick_mfma(D, f->mf_W, cp != 0, f->mf[cp].nl_fixed);
    static const size_t lds_pad = getenv("LSDR_MFMA_SLDS") ? (size_t)atoi(getenv("LSDR_MFMA_SLDS")) : 0;   // tuning hook: extra LDS per stream workgroup (bounds the workgroups resident per CU)
    const size_t lds_bytes = stream ? stream_lds(D, f->bk[cp].nq, cp != 0, snp, sk.fold, false) + lds_pad : blk ? f->bk[cp].lds : f->mf[cp].lds;
    if (lds_bytes > 64 * 1024)
      LSDR_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    unsigned grid = a.tiles_per_xcd * 8;
    // (complex taps on the register-staged kernels: three workgroups per CU queued — two resident ones that start together stay in
    // step and leave the memory idle while both compute: 0.345 ms per 64 Mi against 0.157)
    static const bool wpc_env = getenv("LSDR_MFMA_WPC") != nullptr;
    const int wpc = stream ? (carry ? f->stream_wpc_carry : f->stream_wpc) : (!wpc_env && cp && f->mf_W == 2 ? 3 : f->mf_wpc);
    const unsigned pg = (unsigned)(f->ctx->num_cu * wpc + 7) / 8 * 8;
    if (grid > pg) grid = pg;
    {
      static const bool dbg = getenv("LSDR_FIR_DEBUG") != nullptr;   // diagnostic: what the runtime will co-schedule
      if (dbg) {
        int nb = -1;
        hipFuncAttributes fa;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k, stream ? 64 : 64 * f->mf_W, lds_bytes);
        (void)hipFuncGetAttributes(&fa, (const void *)k);
        fprintf(stderr, "fir_filter: %s cp=%d grid %u x %d lanes, LDS %zu B, regs %d, scratch %zu B, workgroups per CU (occupancy API) %d\n",
                stream ? "k_fir_mfma_stream" : blk ? "k_fir_mfma_blk" : "k_fir_mfma", cp, grid, stream ? 64 : 64 * f->mf_W, lds_bytes,
                fa.numRegs, (size_t)fa.localSizeBytes, nb);
      }
    }
    static const bool skip = LSDR_MEASURE_ENV("LSDR_FIR_SKIP") != nullptr;   // measure build only: after 16 launches, no filter kernel (stale outputs)
    static int launches = 0;
    if (!(skip && ++launches > 16))
      hipLaunchKernelGGL(k, dim3(grid), dim3(stream ? 64 : 64 * f->mf_W), lds_bytes, f->ctx->stream, a);
    LSDR_HIP(hipGetLastError());
    *produced = count;
    *consumed = count * D;
    return LSDR_OK;
  }
  const bool real_path = real_taps;
  int mode = (f->cfg.arith != LSDR_FIR_EXACT ? 2 : 0) + (real_path ? 1 : 0);
  int Rs = 0;
  fir_kernel_t k;
  if (f->spec)
    k = f->cfg.in_format == LSDR_IN_CU8 ? pick_spec<LSDR_IN_CU8>(D, mode, &Rs, f->persist) : pick_spec<LSDR_IN_CF32>(D, mode, &Rs, f->persist);
  else
    k = f->cfg.in_format == LSDR_IN_CU8 ? pick_generic<LSDR_IN_CU8>(f->R, mode) : pick_generic<LSDR_IN_CF32>(f->R, mode);
  if (f->lds_bytes > 64 * 1024)
    LSDR_HIP(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f->lds_bytes));
  unsigned grid = a.tiles_per_xcd * 8;
  if (f->spec && f->persist && grid > f->persist_grid) grid = f->persist_grid;
  hipLaunchKernelGGL(k, dim3(grid), dim3(kThreads), f->lds_bytes, f->ctx->stream, a);
  LSDR_HIP(hipGetLastError());
  *produced = count;
  *consumed = count * D;
  return LSDR_OK;
}

int lsdr_fir_filter_run(lsdr_fir_filter *f, const void *in, size_t n_in, lsdr_cf32 *out, size_t cap_out,
                        size_t *consumed, size_t *produced) {
  return fir_run_streams(f, 1, &in, n_in, &out, cap_out, consumed, produced);
}

int lsdr_fir_filter_run_multi(lsdr_fir_filter *f, unsigned n_streams, const void *const *ins, size_t n_in, lsdr_cf32 *const *outs,
                              size_t cap_out, size_t *consumed, size_t *produced) {
  return fir_run_streams(f, n_streams, ins, n_in, outs, cap_out, consumed, produced);
}

}  // extern "C"
