// leansdr_amd/csrc/fir_stream.h — k_fir_mfma_stream (LSDR_FIR_MFMA_BLK's kernel, fir_filter<cf32,float>::run of dsp.h:233-280 on the
// matrix pipe) and what its launches share: the argument block of every fir_filter kernel, the vector types, the LDS budget.
// Included by fir_filter.hip (the benchmark geometry's compile-time forms, decimations 10 and 30) and by fir_stream_sweep.hip
// (one translation unit per eighth of the decimations 1 … 64: every Fs/(4·Fm) leandvb.cc:353-378 can ask for).
#pragma once
#include "lsdr_internal.h"

namespace lsdr_fir {

struct fir_args {
  const void *in;        // cf32 or cu8 samples
  float2 *out;
  const float2 *sc;      // shifted coefficients [N]   (complex kernels)
  const float *rc;       // real coefficients   [N]   (real kernel)
  const float2 *scp;     // zero-padded to ncols·D taps (persistent kernels)
  const float *rcp;
  unsigned ncols;
  unsigned N, D;
  unsigned S;            // LDS row stride in samples (odd)
  unsigned long long count;      // outputs to produce
  unsigned long long n_in;       // input samples available
  unsigned n_tiles, tiles_per_xcd;
  // lsdr_fir_filter_run_multi (persistent kernels): the same filter over n_streams equal-length buffers in one launch;
  // global tile g = stream·tiles_per_stream + local tile.  n_streams = 1: in/out above.
  unsigned n_streams, tiles_per_stream;
  const void *ins[8];
  float2 *outs[8];
  float in_scale;        // 1.0f → none
  // k_fir_mfma: coefficient operand table (zero-padded, lsdr_fir_filter::d_atab), its length, blocks of four MFMA steps
  const float *mf_atab;
  unsigned mf_alen, mf_blocks;
  unsigned long long *trace;   // LSDR_FIR_TRACE builds: per-wave phase cycle sums
  // k_fir_mfma_stream<…, IV = 1> (the fused auto_notch + fir_filter of notch.hip): the coefficient operand changes along the
  // stream — mf_atab holds n_iv tables of KS·64 floats, table i serves the tiles from iv_tile_first[i] on (ascending, [0] = 0)
  const unsigned *iv_tile_first;
  unsigned n_iv;
  unsigned xcd_rot;            // k_fir_mfma_stream: XCD x starts its walk x·xcd_rot tiles into its range (wrapping): see lsdr_fir_filter::stream_xrot
  unsigned chunked;            // k_fir_mfma_stream: a workgroup's tiles are CONSECUTIVE (one stretch of its XCD's range) instead of strided
};
typedef void (*fir_kernel_t)(fir_args);

// cache policy of the streaming sample loads (buffer_load aux bits on gfx94x/gfx950: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef LSDR_FIR_LOAD_AUX
#define LSDR_FIR_LOAD_AUX 2
#endif
typedef float lsdr_v2f __attribute__((ext_vector_type(2)));
typedef unsigned lsdr_v2u __attribute__((ext_vector_type(2)));
typedef float lsdr_v4f __attribute__((ext_vector_type(4)));
typedef unsigned lsdr_v4u __attribute__((ext_vector_type(4)));
}  // namespace lsdr_fir

// The kernels: one instance per translation unit that names it, under that unit's own namespace name (LSDR_STREAM_NS) — external
// linkage with a unique name (an unnamed namespace in a HEADER leaves hipcc's host-side kernel handles undefined).
#ifndef LSDR_STREAM_NS
#error "define LSDR_STREAM_NS (a namespace name unique to the translation unit) before including fir_stream.h"
#endif
namespace LSDR_STREAM_NS {
using namespace lsdr_fir;

// ---- k_fir_mfma_stream: LSDR_FIR_MFMA_BLK without a staging phase ---------------------------------------------
// k_fir_mfma_blk's trace: the MFMAs themselves are 43 % of a tile; the rest is the wave standing in the vector-memory queue
// with the next tile's loads (in-order issue: a load that cannot be queued blocks the MFMAs behind it), then waiting for them,
// writing 64 KB from registers to LDS, and two barriers.  None of that is needed.  A wavefront walks its 128 rows in order —
// once the sample operands of a pair of row tiles are in registers those 16 rows of LDS are dead — so the NEXT tile's rows are
// loaded straight into them (buffer_load … lds: no registers, no LDS-write instructions, the wave never waits for the queue's
// data, only for its slots) while the wave goes on with the following pairs; they are read again seven pairs later.
// Every wavefront is its own workgroup with a private region (128 rows + the K padding's read-ahead, natural sample order)
// and a private Z ring: no barrier anywhere.  Samples reach the MFMA untouched (the fused scaler rides on the taps, as in
// k_fir_mfma_blk), so the arithmetic is k_fir_mfma_blk's, bit for bit.
// Row stride: the sample operand's ds_read_b32 is conflict-free when a row of D samples is ≡ 4 (mod 8) floats long (32 lanes = 8 rows ×
// {re, im} × 2 K slots on 32 banks).  D ≡ 2 (mod 4) — 10, 30 — is that by itself: the region is the stream's own byte order and an
// LDS-direct load writes 1 KiB of consecutive samples.  Every other D gets PADF = stream_padf(D) floats behind each row (2, 4 or 6: odd D
// too — a padded row starts on a 16-byte granule whatever D is): an LDS-direct load still writes 64 consecutive granules, but the lane
// of granule p of row ρ fetches stream bytes 16·(granule) − 4·PADF·ρ — the padding is filled with the samples that follow the row
// (finite, and only ever multiplied by the K padding's zero taps).  Same arithmetic, bit for bit, for every D.
typedef __attribute__((address_space(3))) void *fir_lds_ptr;
// (128-row tiles could carry too — their ring keeps rows 112 … 127 in slots 48 … 63 until pair 3 is written — and were measured: the headline's launch alone
// 5.37 → 5.38 TB/s, in its pipeline 704–706 → 697–702 GS/s: its halo rows come out of L2 and the launch is HBM-bound; the warm-up pair is all it adds)
#ifndef LSDR_STREAM_CARRY8
#define LSDR_STREAM_CARRY8 0
#endif
constexpr bool stream_carry(int iv, int np, bool fold) { return !iv && ((LSDR_STREAM_CARRY8 && np == 8) || (np == 4 && fold)); }
constexpr unsigned stream_padf(unsigned D) { return (12u - (2u * D) % 8u) % 8u; }      // (2·D + PADF) ≡ 4 (mod 8), PADF < 8

// IV = 1: the taps are a function of the position in the stream (fir_args::iv_tile_first): a wavefront walks its tiles in
// ascending order, so it reloads the coefficient operand the (few) times it crosses into another interval — behind a vmcnt(0),
// so that the hand-counted waits below never see these loads.  One stream per launch.
// The IV launch is OVERSUBSCRIBED (64 workgroups per CU queued, each with a short tile list): with four 39 KB workgroups per CU a grid of
// exactly the resident workgroups is only resident in full while nothing else holds LDS — next to cstln_receiver's staged tiles (9 KB
// per wavefront) some workgroups started when others ENDED and the launch took 1.19 ms instead of 0.76 (256 Mi samples); with 16–32 per CU
// the dispatcher deals the work: 0.80 ms.  (Tiles dealt by a per-XCD atomic counter, one returned atomic per tile: 0.96 ms alone — dropped.)
// NP = pairs of row tiles per wave tile (16 rows each): 8 → a region of 128 rows (38–39 KB of LDS with its ring: four wavefronts per CU, 117
// outputs per 128 rows at 12 tap blocks) — real taps, HBM-bound; 4 → 64 rows, 53 outputs per 64 rows, ring folded to 48 rows (FOLD below):
// 19.6–20.4 KB, EIGHT per CU — complex taps and the IV pass, whose matrix side and memory side take about the same time (0.40 / 0.39–0.44 ms per
// 256 Mi samples alone, 0.48–0.50 together: profiles/r06_bench/fir_bound.txt) and overlap better from two wavefronts per SIMD; 6 → 96 rows (30 KB,
// five per CU): 0.506.  The outputs do not depend on it.
// FOLDT (NP = 4 only): the folded 48-row ring below — its mirrored rows lie over the region's last sample rows, which must be at least
// as many bytes (decimation 30: yes; the small decimations of the sweep: no — they run NP = 8, the large ones NP = 4 unfolded).
template <int DT, int CP, int NQT, int IV = 0, int NP = 8, bool FOLDT = (NP == 4)>
__global__ __launch_bounds__(64) void k_fir_mfma_stream(fir_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr unsigned D = DT;
  constexpr unsigned PADF = stream_padf(D);
  constexpr unsigned KP = (D + 3) / 4 * 4, KS = KP / 4;           // K slots = the samples of a row, four per MFMA step (both tap kinds)
  constexpr unsigned NB = CP ? 3 * KS : KS;                       // coefficient operand registers (CP: re, −im, +im parts — see the lane cursors)
  constexpr unsigned FP = ((KP - D) + 1) & ~1u;                   // samples in front of row 0 that the K padding reads
  constexpr unsigned ROWB = D * 8 + PADF * 4;                     // bytes per row
  constexpr unsigned PAIRG = 16 * ROWB / 16;                      // 16-byte granules per pair of row tiles (16 rows)
  static_assert(ROWB % 16 == 0 && (ROWB / 4) % 8 == 4, "a row starts on a granule and is ≡ 4 (mod 8) floats long");
  constexpr unsigned RW = 16 * NP;                                // rows (blocks of D samples) per wave tile
  static_assert(NP >= 4 && NP % 2 == 0, "whole diagonal batches of two pairs; the wait counts assume NP >= 3");
  static_assert(!FOLDT || (NP == 4 && (stream_carry(IV, NP, FOLDT) || 16 * ROWB >= 16 * 2 * 17 * 4)), "the folded ring's mirrored rows must fit in the last pair's sample rows");
  constexpr unsigned REGB = FP * 8 + RW * ROWB;                   // region bytes
  // LDS-direct loads per refill group.  PADF = 0: group 0 also fetches the FP samples in front of row 0.  PADF > 0: all groups alike —
  // the front is never fetched (it is zeroed once: what the K padding of row 0 reads there only has to be finite).
  constexpr unsigned NLI = PADF ? (PAIRG + 63) / 64 : (PAIRG + FP / 2 + 63) / 64;
  const unsigned l = threadIdx.x;
  // CARRY (the folded 64-row tiles: complex taps at the C2 geometry): the Z rows that the first NQ − 1 outputs of a tile reach back into are the
  // LAST rows of the tile before it — and a workgroup walks consecutive tiles, so they can stay in its ring: tiles are RW rows apart (not
  // RW − (NQ − 1): no halo re-read, no halo MFMAs — 64 rows per 64 outputs instead of per 54) and the first tile of a workgroup's list warms the
  // ring up with the last pair of the tile in front of it (below).  The ring keeps those rows in a GAP of NQ − 1 rows in front of slot 0 (see
  // FOLD) until the next tile's first diagonal batch has read them.  c2_offset 537 → 575–587 GS/s.  The IV pass keeps the halo (its intervals
  // are cut by tiles of MW outputs: notch.hip); so do the 128-row tiles (stream_carry).
  constexpr bool CARRY = stream_carry(IV, NP, FOLDT);
  const unsigned NQ = NQT ? (unsigned)NQT : a.mf_blocks, MW = CARRY ? RW : RW - (NQ - 1);
  constexpr int NQR = NQT ? NQT : 16;
  const unsigned ROWZ = 2 * (NQ | 1u);
  // 64 rows + rows 0…15 once more as rows 64…79 (see k_fir_mfma_blk); folded ring with CARRY: NQ − 1 rows in front of slot 0 (the gap)
  char *const ring = smem_raw + ((REGB + 15) & ~15u) + ((CARRY && FOLDT) ? (NQ - 1) * ROWZ * 4 : 0u);
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  // this XCD's tiles [xcd·tiles_per_xcd, + xcnt), walked from tile x·xcd_rot of the range on, wrapping (xcd_rot = 0: from its start)
  const unsigned xbase = xcd * a.tiles_per_xcd;
  const unsigned xcnt = xbase >= a.n_tiles ? 0u : (a.n_tiles - xbase < a.tiles_per_xcd ? a.n_tiles - xbase : a.tiles_per_xcd);
  const bool chunked = CARRY || a.chunked;      // (CARRY: consecutive tiles, walked from the range's start)
  const unsigned xrot = (!CARRY && xcnt) ? (unsigned)(((unsigned long long)xcd * a.xcd_rot) % xcnt) : 0u;
  auto tile_of = [&](unsigned ti) { const unsigned p = ti + xrot; return xbase + (p >= xcnt ? p - xcnt : p); };
  // a workgroup's tile list: strided (slot, slot + slots, …: at any moment the XCD's workgroups read one narrow window of its range) or one
  // consecutive stretch of ⌈xcnt/slots⌉ tiles (a workgroup stays inside one or two 2 MiB pages: see lsdr_fir_filter::stream_chunked)
  const unsigned per = chunked ? (xcnt + slots - 1) / slots : 0u;
  const unsigned t_first = chunked ? slot * per : slot, t_step = chunked ? 1u : slots;
  const unsigned t_lim = chunked ? (t_first + per < xcnt ? t_first + per : xcnt) : xcnt;
  auto valid = [&](unsigned ti) { return ti < t_lim; };

  float bco[NB];
  unsigned iv_cur = 0, iv_lo = 0, iv_hi = ~0u;      // IV: the interval of the current tile and the tiles [iv_lo, iv_hi) it serves
  if (!IV) {
#pragma unroll
    for (unsigned s = 0; s < NB; ++s) bco[s] = a.mf_atab[s * 64 + l];
  }

  // The region of wave tile `lt` of a stream: row ρ = output-row p = lt·MW − (NQ−1) + ρ, i.e. samples
  // x[N + D·p − (D−1) … N + D·p]; region sample 0 is x[X0], X0 = N + 1 − D·NQ − FP + D·MW·lt (negative at the stream start:
  // those samples meet zero taps — the clamped resource makes their offsets wrap out of range: zeros).
  // One stream per launch (everything but lsdr_fir_filter_run_multi): its pointers are read ONCE — a scalar load from the argument block
  // per tile, with the lgkmcnt(0) it needs in front of the tile's first LDS reads, stalled the wavefront for the load's latency every tile
  // (the IV pass lost 10 % to two such loads per tile: profiles/r06_bench/README.md)
  const bool one = a.n_streams == 1;
  const char *const in0 = reinterpret_cast<const char *>(a.ins[0]);
  float *const out0 = reinterpret_cast<float *>(a.outs[0]);
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned adj = 0;
  auto aim = [&](unsigned tile, bool live) {
    const unsigned st = live && !one ? tile / a.tiles_per_stream : 0u, lt = tile - st * a.tiles_per_stream;
    const long long j0 = (long long)a.N + 1 - (long long)(D * NQ) - (long long)(PADF ? 0u : FP) + (long long)lt * MW * D;   // PADF: row 0's first sample
    const long long jb = j0 < 0 ? 0 : j0;
    const unsigned long long bytes = live ? (a.n_in - (unsigned long long)jb) * 8ull : 0ull;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(one ? in0 : reinterpret_cast<const char *>(a.ins[st])) + (live ? jb * 8 : 0), 0,
                                             (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
    adj = (unsigned)((jb - j0) * 8);
  };
  // padded rows: the stream byte offset (from row 0's first sample of a pair of row tiles) that this lane's granule of load i holds —
  // granule p of the pair = row p / (ROWB/16), so 16·p less the padding of the rows in front of it
  unsigned poff[NLI];
  if constexpr (PADF != 0) {
#pragma unroll
    for (unsigned i = 0; i < NLI; ++i) { const unsigned pg = 64 * i + l; poff[i] = 16u * pg - 4u * PADF * (pg / (ROWB / 16)); }
    if (l < FP * 2) reinterpret_cast<float *>(smem_raw)[l] = 0.f;      // the front: zero, once
  }
  // refill group P = the granules pair P reads first (its rows; pair 0 also the front padding)
  auto refill = [&](int P) {
    if constexpr (PADF != 0) {
      const unsigned base = (unsigned)P * (16u * D * 8u) - adj;      // the pair's first sample, bytes from the resource's start
#pragma unroll
      for (unsigned i = 0; i < NLI; ++i) {
        const unsigned voff = poff[i] + base;      // (a local: with poff[i] inside the builtin's argument list hipcc's HOST pass drops the kernel's definition without a word)
#ifdef LSDR_STREAM_NOLOAD
        if (l == 0xffffffffu)
#else
        if (64 * i + 64 <= PAIRG || 64 * i + l < PAIRG)
#endif
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (fir_lds_ptr)(size_t)(unsigned)(unsigned long long)(smem_raw + FP * 8 + 16 * ((unsigned)P * PAIRG + 64 * i)), 16,
                                                   voff, 0, 0, LSDR_FIR_LOAD_AUX);
      }
      return;
    }
    const unsigned g0 = P ? FP / 2 + (unsigned)P * PAIRG : 0u, g1 = FP / 2 + (unsigned)(P + 1) * PAIRG;
#pragma unroll
    for (unsigned i = 0; i < NLI; ++i) {
      const unsigned g = g0 + 64 * i + l;
#ifdef LSDR_STREAM_NOLOAD      // measurement build: the compute side alone (results are garbage)
      if (g == 0xffffffffu)
#else
      if (g0 + 64 * i + 64 <= g1 || g < g1)      // (whole loads: no lane mask)
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (fir_lds_ptr)(size_t)(unsigned)(unsigned long long)(smem_raw + 16 * (g0 + 64 * i)), 16,
                                                 16 * g - adj, 0, 0, LSDR_FIR_LOAD_AUX);
    }
  };

  unsigned ti = t_first;
  if (!valid(ti)) return;
#ifdef LSDR_STREAM_PRIO
  __builtin_amdgcn_s_setprio(LSDR_STREAM_PRIO);
#endif
  if (IV) {      // the interval of the first tile
    const unsigned t0 = tile_of(ti);
    while (iv_cur + 1 < a.n_iv && t0 >= a.iv_tile_first[iv_cur + 1]) ++iv_cur;
    iv_lo = a.iv_tile_first[iv_cur]; iv_hi = iv_cur + 1 < a.n_iv ? a.iv_tile_first[iv_cur + 1] : ~0u;
#pragma unroll
    for (unsigned s = 0; s < NB; ++s) bco[s] = a.mf_atab[(size_t)iv_cur * (NB * 64) + s * 64 + l];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // per-lane cursors (see k_fir_mfma_blk); the sample operand of K slot r' = 4·s + k of row ρ is the sample r' BEFORE the row's last one.
  // Real taps (CP = 0): the 16 operand rows of an MFMA are 8 rows × {re, im} — a pair of row tiles is two MFMA row tiles (h), one
  // 4-byte operand per lane and step, Z[(ρ, comp)][q] += x_comp · c.
  // Complex taps (CP = 1): the 16 operand rows are the pair's 16 rows and a lane reads BOTH components of its sample in one 8-byte
  // ds_read (32 lanes × 8 B on 64 banks: conflict-free under the same row rule); two accumulators, four MFMAs per step:
  //   Zre += xr·cr,  Zim += xi·cr,  Zre += xi·(−ci),  Zim += xr·(+ci)
  // — the operands reach the matrix pipe as loaded (no sign flips on the vector ALUs: a VALU result in front of every MFMA cost 40 % of
  // the pipe, profiles/r06_bench/fir_bound.txt), the signs live in the coefficient operand (bco[KS + s] = −ci, bco[2·KS + s] = +ci).
  // Per block and component that is the chain lo_fir_filter_blk states: groups of four taps, the re-part products of a group, then
  // its im-part products.
  const unsigned kq = l >> 4 & 3u, i16 = l & 15u, beta = CP ? i16 : i16 >> 1, c = CP ? 0u : i16 & 1u;
  constexpr unsigned ASTEP = 32, ATILE = 8 * ROWB;
  // float index of (row ρ, step s): 2·(FP + D·ρ + D − 1 − r') + comp, r' = 4·s + k  →  lane part at s = KS−1
  const int rlast = 4 * (int)(KS - 1) + (int)kq;
  const unsigned a0 = (unsigned)(2 * ((int)FP + (int)D - 1 - rlast) + (int)c) * 4u + beta * ROWB;   // bytes, ≥ 0 by FP
  const unsigned zq = l & 15u, zrow = 2 * kq;
  const unsigned ro = l >> 1, rc = l & 1u;
  // diagonal sum (rows ascend with the output index here): output of row ρ = 32·B + o is Σ_q Z[ρ − q][q]; its row as 16 … 79
  // (rows 0…15 are mirrored at 64…79) so that ρ − q needs no wrap; byte offset of term q = 15 for even / odd B, terms with
  // smaller q `dstep` bytes further on
  const unsigned dstep = ROWZ * 4 - 8;
  // NP = 4 (FOLD): a wave tile is 64 ring rows and nothing wraps except the reads of rows 0 … NQ−2, which are not outputs — no mirrored rows
  // 64 … 79.  And the ring is FOLDED to 48 rows: rows 0 … 15 (pair 0) are dead once the first diagonal batch has read them (during pair 2), so
  // pair 3's rows 48 … 63 go there; the rows in front of them that the second batch's sums reach back into (37 … 47, pair 2) are written a
  // second time at rows −16 … −1, i.e. over the LAST sample rows of the region — pair 3's, whose operands are in registers by then and whose
  // refill is issued after that batch's reads.  20 KB of LDS per wavefront instead of 21–22: EIGHT per CU.
  constexpr bool MIRROR = NP != 4, FOLD = FOLDT;
  int dbase[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const unsigned rr0 = 32u * par + ro, rr = (FOLD && rr0 >= 48) ? rr0 - 48 : rr0, rp = (MIRROR && rr < 16) ? rr + 64 : rr;
    dbase[par] = (int)((rp * ROWZ + rc) * 4) - 15 * (int)dstep;   // = ((rp − 15)·ROWZ + 2·15 + rc)·4
  }

  const char *ap = smem_raw + a0;
  unsigned pa[2][2][CP ? 1 : KS];
  lsdr_v2f pc[2][CP ? KS : 1];
  auto fetch1 = [&](int set, int pair, int h, unsigned s) {
    pa[set][h][s] = *reinterpret_cast<const unsigned *>(ap + (2 * pair + h) * ATILE + (KS - 1 - s) * ASTEP);
  };
  auto fetch2 = [&](int set, int pair, unsigned s) {      // CP: (re, im) of the lane's sample of step s
    pc[set][s] = *reinterpret_cast<const lsdr_v2f *>(ap + 2 * pair * ATILE + (KS - 1 - s) * ASTEP);
  };
  auto opnd = [&](unsigned r) { return __uint_as_float(r); };
  lsdr_v4f acc[2][2];      // CP = 0: [set][h];  CP = 1: [set][re / im]
  auto to_ring = [&](int set, int pair) {
    if (CP) {
      if (zq < NQ) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {      // the accumulators' rows 4·kq + i of the pair, column zq: (re, im) side by side as the ring holds them
          const unsigned row0 = (16u * pair + 4u * kq + i) & 63u, row = (FOLD && pair == 3) ? row0 - 48u : row0;
          const lsdr_v2f v = {acc[set][0][i], acc[set][1][i]};
          *reinterpret_cast<lsdr_v2f *>(ring + (row * ROWZ + 2 * zq) * 4) = v;
          if (FOLD && pair == 2 && (!CARRY || 4u * kq + i + NQ >= 17u)) *reinterpret_cast<lsdr_v2f *>(ring + (((int)row - 48) * (int)ROWZ + 2 * (int)zq) * 4) = v;
          if (MIRROR && ((16u * pair) & 63u) == 0) *reinterpret_cast<lsdr_v2f *>(ring + ((row + 64) * ROWZ + 2 * zq) * 4) = v;
        }
      }
      return;
    }
    if (zq < NQ) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned row0 = (16u * pair + 8u * h + zrow) & 63u, row = (FOLD && pair == 3) ? row0 - 48u : row0;
        const lsdr_v2f lo = {acc[set][h][0], acc[set][h][1]}, hi = {acc[set][h][2], acc[set][h][3]};
        *reinterpret_cast<lsdr_v2f *>(ring + (row * ROWZ + 2 * zq) * 4) = lo;
        *reinterpret_cast<lsdr_v2f *>(ring + ((row + 1) * ROWZ + 2 * zq) * 4) = hi;
        if (FOLD && pair == 2) {            // rows 32 … 47 once more at −16 … −1 (!CARRY: inside the region's last rows; CARRY: the rows the gap has — see FOLD)
          if (!CARRY || 8u * h + zrow + NQ >= 17u) *reinterpret_cast<lsdr_v2f *>(ring + (((int)row - 48) * (int)ROWZ + 2 * (int)zq) * 4) = lo;
          if (!CARRY || 8u * h + zrow + 1u + NQ >= 17u) *reinterpret_cast<lsdr_v2f *>(ring + (((int)row - 47) * (int)ROWZ + 2 * (int)zq) * 4) = hi;
        }
        if (MIRROR && ((16u * pair) & 63u) == 0) {
          *reinterpret_cast<lsdr_v2f *>(ring + ((row + 64) * ROWZ + 2 * zq) * 4) = lo;
          *reinterpret_cast<lsdr_v2f *>(ring + ((row + 65) * ROWZ + 2 * zq) * 4) = hi;
        }
      }
    }
  };
  // FOLD with CARRY: the last NQ − 1 rows of the tile (pair 3's) once more in the gap in front of slot 0, where the NEXT tile's first diagonal
  // batch finds them — written after this tile's last batch has read the gap (it held rows 38 … 47 for it)
  auto carry_write = [&](int set) {
    if (zq < NQ) {
      if (CP) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4u * kq + i + NQ >= 17u) {
            const lsdr_v2f v = {acc[set][0][i], acc[set][1][i]};
            *reinterpret_cast<lsdr_v2f *>(ring + (((int)(4u * kq + i) - 16) * (int)ROWZ + 2 * (int)zq) * 4) = v;
          }
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const lsdr_v2f lo = {acc[set][h][0], acc[set][h][1]}, hi = {acc[set][h][2], acc[set][h][3]};
          if (8u * h + zrow + NQ >= 17u) *reinterpret_cast<lsdr_v2f *>(ring + (((int)(8u * h + zrow) - 16) * (int)ROWZ + 2 * (int)zq) * 4) = lo;
          if (8u * h + zrow + 1u + NQ >= 17u) *reinterpret_cast<lsdr_v2f *>(ring + (((int)(8u * h + zrow) - 15) * (int)ROWZ + 2 * (int)zq) * 4) = hi;
        }
      }
    }
  };

  // Start: the first tile's rows on their way.  CARRY, first tile of the list not the first of its stream: the ring has to hold the last
  // NQ − 1 rows of the tile in FRONT of it — its last pair is loaded (with the pair before it: the K padding reads that one's last samples),
  // multiplied and written to the ring while the first tile's other rows are already coming; then its two places are refilled in order,
  // so that the loop below finds the NP groups of its first tile outstanding, oldest first, as without the warm-up.
  {
    const unsigned t0 = tile_of(ti);
    const unsigned lt0 = one ? t0 : t0 % a.tiles_per_stream;
    if (CARRY && lt0 > 0) {
      aim(t0 - 1, true);
      refill(NP - 2); refill(NP - 1);
      aim(t0, true);
#pragma unroll
      for (int P = 0; P < NP - 2; ++P) refill(P);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NP - 2) * NLI) : "memory");
      if (CP) {
#pragma unroll
        for (unsigned s = 0; s < KS; ++s) fetch2(0, NP - 1, s);
#pragma unroll
        for (unsigned s = 0; s < KS; ++s) {
          const float xr = pc[0][s][0], xi = pc[0][s][1];
          const lsdr_v4f z0 = s ? acc[0][0] : (lsdr_v4f){0.f, 0.f, 0.f, 0.f}, z1 = s ? acc[0][1] : (lsdr_v4f){0.f, 0.f, 0.f, 0.f};
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr, bco[s], z0, 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xi, bco[s], z1, 0, 0, 0);
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xi, bco[KS + s], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr, bco[2 * KS + s], acc[0][1], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (unsigned s = 0; s < KS; ++s) { fetch1(0, NP - 1, 0, s); fetch1(0, NP - 1, 1, s); }
#pragma unroll
        for (unsigned s = 0; s < KS; ++s) {
          const lsdr_v4f z0 = s ? acc[0][0] : (lsdr_v4f){0.f, 0.f, 0.f, 0.f}, z1 = s ? acc[0][1] : (lsdr_v4f){0.f, 0.f, 0.f, 0.f};
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(opnd(pa[0][0][s]), bco[s], z0, 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(opnd(pa[0][1][s]), bco[s], z1, 0, 0, 0);
        }
      }
      to_ring(0, NP - 1);
      if (FOLD) carry_write(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      refill(NP - 2); refill(NP - 1);
    } else {
      aim(t0, true);
#pragma unroll
      for (int P = 0; P < NP; ++P) refill(P);
    }
  }

  while (true) {
    const unsigned tile = tile_of(ti);
    const unsigned tn = ti + t_step;
    const bool more = valid(tn);
    const unsigned st = one ? 0u : tile / a.tiles_per_stream;
    const unsigned lt = tile - st * a.tiles_per_stream;
    const unsigned long long m0 = (unsigned long long)lt * MW;
    const bool low_rows = CARRY && lt > 0;      // the outputs of rows 0 … NQ − 2 exist: their Z terms of the tile in front are in the ring
    float *const po = one ? out0 : reinterpret_cast<float *>(a.outs[st]);
    aim(more ? tile_of(tn) : 0u, more);      // the refills of this iteration fetch the NEXT tile (an empty resource at the end: no traffic)
    if (IV && (tile >= iv_hi || tile < iv_lo)) {      // (wave-uniform, a handful of times per launch: the bounds of the current interval stay in registers)
      unsigned iv = iv_cur;
      while (iv + 1 < a.n_iv && tile >= a.iv_tile_first[iv + 1]) ++iv;
      while (iv > 0 && tile < a.iv_tile_first[iv]) --iv;       // (the walk wraps once when it does not start at the range's first tile)
      iv_lo = a.iv_tile_first[iv]; iv_hi = iv + 1 < a.n_iv ? a.iv_tile_first[iv + 1] : ~0u;
      if (iv != iv_cur) {
        iv_cur = iv;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (unsigned s = 0; s < NB; ++s) bco[s] = a.mf_atab[(size_t)iv_cur * (NB * 64) + s * 64 + l];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }

    float zv[NQR];
    auto diag_read = [&](int batch, int q) {
      zv[q] = *reinterpret_cast<const float *>(ring + (dbase[batch & 1] + (15 - q) * (int)dstep));
    };
    float ysum = 0.f;
    auto diag_add = [&](int q) {
      const float t = ysum + zv[q];
      ysum = q == 0 ? zv[0] : (NQT || (unsigned)q < NQ ? t : ysum);
    };
    auto diag_store = [&](int batch) {
      const int rho = 32 * batch + (int)ro;
      const unsigned long long m = (unsigned long long)((long long)m0 + (rho - (int)(NQ - 1)));      // (rho < NQ − 1 in the first tile of a stream: wraps past count — not stored)
      if ((rho >= (int)(NQ - 1) || low_rows) && m < a.count)
        asm volatile("global_store_dword %0, %1, off" ::"v"(po + 2 * m + rc), "v"(ysum) : "memory");
    };
    auto at_step = [](int i, int n, int lo, int hi) { return hi > lo ? lo + i * (hi - lo) / n : lo; };

    // Wait counts (vector-memory operations retire in order; the hidden output stores only make the counter larger, i.e. the
    // waits stricter).  Refill group j of the previous iteration must have landed before fetch(j).  Behind it in the queue:
    // the previous iteration's groups j+1 … NP−1 and this iteration's groups issued so far (group P−1 goes out at the END of
    // pair P, behind fetch(P+1)): fetch(0): NP−1 groups; fetch(1), during pair 0: NP−2; fetch(j ≥ 2), during pair j−1: NP−3.
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NP - 1) * NLI) : "memory");
#pragma unroll
    for (unsigned s = 0; s < KS; ++s) { if (CP) fetch2(0, 0, s); else { fetch1(0, 0, 0, s); fetch1(0, 0, 1, s); } }
#pragma unroll
    for (int pair = 0; pair < NP; ++pair) {
      const int set = pair & 1;
      if (pair < NP - 1) {
        if (pair == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NP - 2) * NLI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NP - 3) * NLI) : "memory");
      }
#pragma unroll
      for (unsigned s = 0; s < KS; ++s) {
#ifdef LSDR_STREAM_NOMFMA       // measurement build: the memory side alone (one VALU op stands in for each MFMA)
        if (s == 0) { acc[set][0] = (lsdr_v4f){0.f, 0.f, 0.f, 0.f}; acc[set][1] = acc[set][0]; }
        if (CP) {
          acc[set][0][s & 3] += pc[set][s][0] * bco[s]; acc[set][1][s & 3] += pc[set][s][1] * bco[s];
          acc[set][0][s & 3] += pc[set][s][1] * bco[KS + s]; acc[set][1][s & 3] += pc[set][s][0] * bco[2 * KS + s];
        } else {
          acc[set][0][s & 3] += opnd(pa[set][0][s]) * bco[s];
          acc[set][1][s & 3] += opnd(pa[set][1][s]) * bco[s];
        }
#else
        if (CP) {
          const float xr = pc[set][s][0], xi = pc[set][s][1];
          if (s == 0) {
            acc[set][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr, bco[0], (lsdr_v4f){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            acc[set][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xi, bco[0], (lsdr_v4f){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          } else {
            acc[set][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr, bco[s], acc[set][0], 0, 0, 0);
            acc[set][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xi, bco[s], acc[set][1], 0, 0, 0);
          }
          acc[set][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xi, bco[KS + s], acc[set][0], 0, 0, 0);
          acc[set][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr, bco[2 * KS + s], acc[set][1], 0, 0, 0);
        } else if (s == 0) {
          acc[set][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(opnd(pa[set][0][0]), bco[0], (lsdr_v4f){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          acc[set][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(opnd(pa[set][1][0]), bco[0], (lsdr_v4f){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
          acc[set][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(opnd(pa[set][0][s]), bco[s], acc[set][0], 0, 0, 0);
          acc[set][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(opnd(pa[set][1][s]), bco[s], acc[set][1], 0, 0, 0);
        }
#endif
        if (CP) {
          if (pair < NP - 1) fetch2(set ^ 1, pair + 1, s);
        } else if (pair < NP - 1 && !(s & 1)) {
#pragma unroll
          for (int h = 0; h < 2; ++h) { if (s + 1 < KS) fetch1(set ^ 1, pair + 1, h, s + 1); fetch1(set ^ 1, pair + 1, h, s); }
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
        // rows of pair P−1: every operand of them has been consumed by an MFMA by now (program order) — refill them
        if (pair >= 1 && s == KS - 1) refill(pair - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    to_ring((NP - 1) & 1, NP - 1);
    if (!FOLD) refill(NP - 1);
#pragma unroll
    for (int q = 0; q < NQR; ++q) diag_read(NP / 2 - 1, q);
#pragma unroll
    for (int q = 0; q < NQR; ++q) diag_add(q);
    if (FOLD) {                                // (!CARRY: the last rows' refill lands on the mirrored ring rows: only after the sums have their terms)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (CARRY) carry_write((NP - 1) & 1);
      refill(NP - 1);
    }
    diag_store(NP / 2 - 1);
    if (!more) break;
    ti = tn;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LDS bytes of one k_fir_mfma_stream wavefront: region (front padding + 16·np padded rows) + Z ring + the diagonal reads' overrun
unsigned stream_lds(unsigned D, unsigned nq, bool cplx, unsigned np = 8, bool fold = true, bool iv = false) {
  (void)cplx;      // (the region is the same for both tap kinds: four samples of a row per MFMA step)
  const unsigned kp = (D + 3) / 4 * 4, fp = ((kp - D) + 1) & ~1u;
  const unsigned region = (fp * 8 + 16 * np * (D * 8 + stream_padf(D) * 4) + 15) & ~15u;
  if (np == 4 && fold) return region + (48 + (iv ? 0u : nq - 1)) * 2 * (nq | 1u) * 4;   // folded ring (CARRY: + the gap), compile-time tap blocks only (no overrun)
  return region + (np == 4 ? 64 : 80) * 2 * (nq | 1u) * 4 + 128;
}

// What a launch needs to know about the kernel picked for a geometry: np = pairs of row tiles per wave tile, fold = the folded ring
struct stream_kernel { fir_kernel_t k; unsigned np; bool fold; bool carry() const { return stream_carry(0, (int)np, fold); } };
// The sweep's shape per decimation and tap kind: 128-row wave tiles while five or more wavefronts per CU find their LDS (≤ 32 KB each), 64-row ones
// (unfolded ring) above — real taps D ≤ LSDR_SWEEP_NP8_REAL, complex taps (twice the matrix work per row: more wavefronts to overlap it) D ≤ LSDR_SWEEP_NP8_CPLX
#ifndef LSDR_SWEEP_NP8_REAL
#define LSDR_SWEEP_NP8_REAL 20
#endif
#ifndef LSDR_SWEEP_NP8_CPLX
#define LSDR_SWEEP_NP8_CPLX 12
#endif
constexpr unsigned stream_sweep_np(unsigned D, bool cplx) { return D <= (cplx ? LSDR_SWEEP_NP8_CPLX : LSDR_SWEEP_NP8_REAL) ? 8u : 4u; }
constexpr unsigned kStreamMaxD = 64;
}  // namespace LSDR_STREAM_NS
