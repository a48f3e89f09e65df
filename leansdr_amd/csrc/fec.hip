// leansdr_amd/csrc/fec.hip — DVB-S FEC tail on gfx950 (integer / byte work, bit-exact).
//
//   deconvol_sync   dvb.h:122-513   algebraic deconvolution: every output bit is a parity over a
//                                   64-bit window of I/Q bits → one thread per output byte.
//   mpeg_sync       dvb.h:712-891   bit phase / polarity / sync search + lock tracking: a sequential
//                                   state machine over 204-byte packets; one workgroup, lane 0 keeps
//                                   the reference's state, all lanes move and test bytes.
//   deinterleaver   dvb.h:926-948   fixed gather, one thread per byte.
//   rs_decoder      dvb.h:985-1058 + rs.h   one wavefront per packet: 16 lanes run the syndromes
//                                   (Horner, log/exp tables in LDS), lane 0 runs Berlekamp-Massey /
//                                   Chien / Forney for corrupted packets.
//   derandomizer    dvb.h:1107-1163 PRBS XOR with resync and packet dropping: flags → scan → apply.
//
// These are HBM-streaming byte kernels (no MFMA, nothing to reshape into a GEMM); their rates are
// 1/8…1/16 of the symbol rate, so the design goal is exactness with full parallelism per packet.
#include "lsdr_internal.h"

namespace {

constexpr int kRS = 204, kTS = 188;
constexpr size_t kMsyncSplitPackets = 128;   // locked runs at least this long take the chip-wide realign + bookkeeping split
constexpr int kSync = 0x47, kSyncInv = 0xb8, kCorrupt = 0x55;

__device__ __forceinline__ int par64(unsigned long long x) { return __popcll(x) & 1; }

// ======================================================================== deconvol_sync
struct deconv_dev {
  unsigned long long deconv[8];
  unsigned char lut[4];   // I/Q bits per hard symbol for the locked alignment: lut[symbol] (dvb.h:377)
  int pp, pw;             // punctperiod, punctweight
};
struct deconv_carry {      // sync_t fields that cross run() calls (dvb.h:297-306)
  unsigned long long in;   // shift register of I/Q bits
  unsigned long long out;  // pending decoded bits
};

// Decoding plan of one call, all derived on the host from (n_in0, n_out0, n):
//   refill q (q = 0…R-1) shifts in m(q) symbols and emits pp bits; m(0) = m0, m(q>0) = pw/2.
//   output bit stream = n_out0 carried bits ++ refill bits; byte k = bits [8k, 8k+8).
struct deconv_plan {
  const lsdr_softsymbol *in;       // soft symbols, or (PACKED) null
  const unsigned *in_words;        // PACKED: "hs2" hard symbols, 16 per word MSB first (rx_tiling.h); symbol i of this call is
  unsigned long long in_off;       //         symbol in_off + i of that stream
  unsigned char *out;
  unsigned long long n_bytes;
  unsigned long long refills;      // R
  unsigned m0;                     // symbols shifted in by refill 0
  int n_out0;                      // carried bits in front of the stream
  deconv_carry *carry;             // in: state at call start; out: state at call end
  deconv_carry *carry_next;
  int n_out_end;
};

template <bool PACKED>
__device__ __forceinline__ unsigned deconv_sym(const deconv_plan &P, unsigned long long i) {   // hard symbol i of this call
  if (PACKED) {
    const unsigned long long k = P.in_off + i;
    return (P.in_words[k >> 4] >> (30 - 2 * (int)(k & 15))) & 3u;
  }
  return P.in[i].symbol & 3u;
}
// I/Q window after refill q: the 64 newest I/Q bits, newest symbol in the low bits (dvb.h:378-384).
template <bool PACKED>
__device__ __forceinline__ unsigned long long deconv_window(const deconv_dev &D, const deconv_plan &P,
                                                            unsigned long long in0, unsigned long long q) {
  const unsigned long long nsym = P.m0 + q * (unsigned)(D.pw / 2);   // symbols shifted in so far
  unsigned long long w = 0;
  const unsigned take = nsym < 32 ? (unsigned)nsym : 32u;
  if (PACKED && take == 32) {
    // the 32 symbols [nsym−32, nsym) are 64 consecutive bits of the packed stream, oldest first — already the window's
    // order; the symbol → I/Q map of the locked alignment is applied to all of them at once
    const unsigned long long o = P.in_off + nsym - 32;
    const unsigned long long wi = o >> 4;
    const int sh = 2 * (int)(o & 15);
    const unsigned a = P.in_words[wi], b = P.in_words[wi + 1], c = sh ? P.in_words[wi + 2] : 0u;
    const unsigned hi = sh ? (a << sh) | (b >> (32 - sh)) : a, lo = sh ? (b << sh) | (c >> (32 - sh)) : b;
    const unsigned map4 = (unsigned)D.lut[0] | ((unsigned)D.lut[1] << 2) | ((unsigned)D.lut[2] << 4) | ((unsigned)D.lut[3] << 6);
    const unsigned M = 0x55555555u;
    unsigned res[2];
    const unsigned xs[2] = {hi, lo};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned x = xs[h], b1 = (x >> 1) & M, b0 = x & M;
      const unsigned sel[4] = {~b1 & ~b0 & M, ~b1 & b0, b1 & ~b0, b1 & b0};
      unsigned o2 = 0;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const unsigned m = (map4 >> (2 * v)) & 3u;
        o2 |= (m & 1u ? sel[v] : 0u) | (m & 2u ? sel[v] << 1 : 0u);
      }
      res[h] = o2;
    }
    return ((unsigned long long)res[0] << 32) | res[1];
  }
  for (unsigned k = 0; k < take; ++k)   // oldest of the `take` first
    w = (w << 2) | D.lut[deconv_sym<PACKED>(P, nsym - take + k)];
  if (take < 32) w |= in0 << (2 * take);
  return w;
}

// byte k of a call's output (readbyte, dvb.h:386-394, for every bit of the byte)
template <bool PACKED>
__device__ __forceinline__ unsigned char deconv_byte(const deconv_dev &D, const deconv_plan &P, unsigned long long in0, unsigned long long out0,
                                                     unsigned long long k) {
  unsigned v = 0;
  long long g = (long long)(8 * k) - P.n_out0;   // index into the refill bit stream (negative: carried bits)
  unsigned long long q_cached = ~0ull, w = 0;
  // refill index and bit position of the first refill bit of this byte: ONE division per thread, then counted along
  const long long g0 = g < 0 ? 0 : g;
  unsigned long long q = (unsigned long long)g0 / (unsigned)D.pp;
  int r = (int)((unsigned long long)g0 - q * (unsigned)D.pp);
  for (int b = 0; b < 8; ++b, ++g) {
    unsigned bit;
    if (g < 0) bit = (unsigned)(out0 >> (unsigned)(-g - 1)) & 1u;   // carried bits, MSB first
    else {
      const int bi = D.pp - 1 - r;
      if (q != q_cached) {
        if (q_cached != ~0ull && q == q_cached + 1) {   // slide by one refill
          const unsigned long long nsym = P.m0 + q * (unsigned)(D.pw / 2);
          for (int s = D.pw / 2; s > 0; --s) w = (w << 2) | D.lut[deconv_sym<PACKED>(P, nsym - s)];
        } else w = deconv_window<PACKED>(D, P, in0, q);
        q_cached = q;
      }
      bit = (unsigned)par64(w & D.deconv[bi]);
      if (++r == D.pp) { r = 0; ++q; }
    }
    v = (v << 1) | bit;
  }
  return (unsigned char)v;
}
// sync_t state after the call (dvb.h:297-306): the shift register and the bits still pending
template <bool PACKED>
__device__ __forceinline__ deconv_carry deconv_carry_after(const deconv_dev &D, const deconv_plan &P, unsigned long long in0, unsigned long long out0) {
  deconv_carry c;
  if (P.refills) {
    c.in = deconv_window<PACKED>(D, P, in0, P.refills - 1);
    const unsigned long long w = c.in;
    // bits still pending come from the tail of the last refill(s); n_out_end < pp + 8: rebuilt exactly — they are the last
    // n_out_end bits of the stream
    unsigned long long o = 0;
    for (int t = P.n_out_end; t > 0; --t) {
      const long long g = (long long)(P.refills * (unsigned)D.pp) - t;   // refill-stream index
      unsigned bit;
      if (g < 0) bit = (unsigned)(out0 >> (unsigned)(-g - 1)) & 1u;
      else {
        const unsigned long long q = (unsigned long long)g / (unsigned)D.pp;
        const int bi = D.pp - 1 - (int)((unsigned long long)g % (unsigned)D.pp);
        const unsigned long long wq = q == P.refills - 1 ? w : deconv_window<PACKED>(D, P, in0, q);
        bit = (unsigned)par64(wq & D.deconv[bi]);
      }
      o = (o << 1) | bit;
    }
    c.out = o;
  } else {
    c.in = in0;
    c.out = out0;
  }
  return c;
}

// ---- rate 1/2 on packed symbols, 32 output bits at a time ----------------------------------------------------------------------
// pp = 1, pw = 2 (dvb.h:480-484): every refill shifts ONE symbol in and emits ONE bit, bit = parity(window & deconv[0]) with the
// newest symbol in the window's low bits (dvb.h:378-394) — and the inverse polynomial of the DVB-S code is short (0x3ba: the last five
// symbols).  So 32 consecutive output bits are a bit-parallel expression of the mapped I/Q bit stream X (newest symbol of the LAST of
// the 32 bits at X[1:0]):  Y = XOR over the set bits j of deconv[0] of (X >> j),  bit t (t = 0 oldest) = Y[2·(31 − t)].
// Same bits as deconv_byte (tests/test_gpu_fec.py compares the two paths); used where every symbol the 32 bits need belongs to this call.
__device__ __forceinline__ unsigned deconv_map16(unsigned x, unsigned map4) {      // the alignment's symbol → I/Q map on 16 packed symbols
  const unsigned M = 0x55555555u, b1 = (x >> 1) & M, b0 = x & M;
  const unsigned sel[4] = {~b1 & ~b0 & M, ~b1 & b0, b1 & ~b0, b1 & b0};
  unsigned o = 0;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const unsigned m = (map4 >> (2 * v)) & 3u;
    o |= (m & 1u ? sel[v] : 0u) | (m & 2u ? sel[v] << 1 : 0u);
  }
  return o;
}
__device__ __forceinline__ unsigned deconv_even16(unsigned y) {                     // bits 0, 2, …, 30 of y → bits 0 … 15
  unsigned x = y & 0x55555555u;
  x = (x | (x >> 1)) & 0x33333333u;
  x = (x | (x >> 2)) & 0x0f0f0f0fu;
  x = (x | (x >> 4)) & 0x00ff00ffu;
  x = (x | (x >> 8)) & 0x0000ffffu;
  return x;
}
__device__ __forceinline__ bool deconv_r12_ok(const deconv_dev &D, const deconv_plan &P) {
  return D.pp == 1 && D.pw == 2 && (D.deconv[0] >> 32) == 0 && P.in_words != nullptr;
}
// bytes [k, k + 4) of the call: where their four packed words start (wi − 3 … wi) and the bit offset; false: not eligible here (the caller
// takes deconv_byte)
__device__ __forceinline__ bool deconv_r12_addr(const deconv_plan &P, unsigned long long k, unsigned long long &wi, int &sh) {
  const long long q0 = (long long)(8 * k) - P.n_out0;                  // refill of the first of the 32 bits
  if (q0 < 0 || k + 4 > P.n_bytes) return false;
  const long long i0 = (long long)P.m0 + q0 - 1;                       // its newest symbol (index in this call)
  if (i0 < 16) return false;
  const unsigned long long g31 = P.in_off + (unsigned long long)i0 + 31;   // newest symbol of the last bit, in the packed stream
  wi = g31 >> 4;
  if (wi < 3) return false;
  sh = 2 * (15 - (int)(g31 & 15));
  return true;
}
// … and the 32 bits from the four words W0 (oldest) … W3.  The symbol → I/Q maps of the four alignments (init_syncs, dvb.h:309-366) are
// bit permutations with inversions — rotations and conjugations of the constellation — i.e. on packed symbols "swap the two bits of every
// symbol or not, then XOR a constant": five instructions per word instead of the generic four-way select.  POLY: the inverse polynomial
// as a compile-time constant (0x3ba, DVB-S rate 1/2: seven shifts, no tests), 0 = the run-time one.
__device__ __forceinline__ bool deconv_affine(const unsigned char *lut, bool &swap, unsigned &mask) {
  const unsigned l0 = lut[0], l1 = lut[1], l2 = lut[2], l3 = lut[3];
  swap = ((l0 ^ l1) & 3u) == 2u;
  const unsigned p1 = swap ? 2u : 1u, p2 = swap ? 1u : 2u;
  mask = (l0 & 3u) * 0x55555555u;
  return ((l1 ^ l0) & 3u) == p1 && ((l2 ^ l0) & 3u) == p2 && ((l3 ^ l0) & 3u) == 3u;
}
__device__ __forceinline__ unsigned deconv_map16a(unsigned x, bool swap, unsigned mask) {
  const unsigned y = swap ? ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1) : x;
  return y ^ mask;
}
template <unsigned POLY>
__device__ __forceinline__ unsigned deconv_r12_bits_t(const deconv_dev &D, unsigned W0, unsigned W1, unsigned W2, unsigned W3, int sh) {
  bool swap; unsigned mask;
  unsigned M0, M1, M2, M3;
  if (deconv_affine(D.lut, swap, mask)) {       // (uniform)
    M3 = deconv_map16a(W3, swap, mask); M2 = deconv_map16a(W2, swap, mask); M1 = deconv_map16a(W1, swap, mask); M0 = deconv_map16a(W0, swap, mask);
  } else {
    const unsigned map4 = (unsigned)D.lut[0] | ((unsigned)D.lut[1] << 2) | ((unsigned)D.lut[2] << 4) | ((unsigned)D.lut[3] << 6);
    M3 = deconv_map16(W3, map4); M2 = deconv_map16(W2, map4); M1 = deconv_map16(W1, map4); M0 = deconv_map16(W0, map4);
  }
  const unsigned x0 = __builtin_amdgcn_alignbit(M2, M3, sh), x1 = __builtin_amdgcn_alignbit(M1, M2, sh), x2 = __builtin_amdgcn_alignbit(M0, M1, sh);
  unsigned y0 = 0, y1 = 0;
  const unsigned poly = POLY ? POLY : (unsigned)D.deconv[0];
#pragma unroll
  for (int j = 0; j < 32; ++j)
    if ((poly >> j) & 1u) { y0 ^= __builtin_amdgcn_alignbit(x1, x0, j); y1 ^= __builtin_amdgcn_alignbit(x2, x1, j); }
  return (deconv_even16(y1) << 16) | deconv_even16(y0);
}
__device__ __forceinline__ unsigned deconv_r12_bits(const deconv_dev &D, unsigned W0, unsigned W1, unsigned W2, unsigned W3, int sh) {
  return (unsigned)D.deconv[0] == 0x3bau ? deconv_r12_bits_t<0x3bau>(D, W0, W1, W2, W3, sh) : deconv_r12_bits_t<0u>(D, W0, W1, W2, W3, sh);
}
__device__ __forceinline__ bool deconv_r12_word(const deconv_dev &D, const deconv_plan &P, unsigned long long k, unsigned &r) {
  unsigned long long wi; int sh;
  if (!deconv_r12_addr(P, k, wi, sh)) return false;
  r = deconv_r12_bits(D, P.in_words[wi - 3], P.in_words[wi - 2], P.in_words[wi - 1], P.in_words[wi], sh);
  return true;
}
__device__ __forceinline__ void deconv_store4(unsigned char *o, unsigned r) {
  if (((unsigned long long)o & 3ull) == 0) *reinterpret_cast<unsigned *>(o) = __builtin_bswap32(r);      // (uniform: the call's output is 4-byte aligned or not)
  else { o[0] = (unsigned char)(r >> 24); o[1] = (unsigned char)(r >> 16); o[2] = (unsigned char)(r >> 8); o[3] = (unsigned char)r; }
}
// the bytes [4·g, 4·g + 4) ∩ [0, n_bytes) of a call
template <bool PACKED>
__device__ __forceinline__ void deconv_group4(const deconv_dev &D, const deconv_plan &P, unsigned long long in0, unsigned long long out0,
                                              unsigned long long g, bool r12) {
  const unsigned long long k = 4 * g;
  unsigned r;
  if (PACKED && r12 && deconv_r12_word(D, P, k, r)) { deconv_store4(P.out + k, r); return; }
  for (unsigned long long b = k; b < k + 4 && b < P.n_bytes; ++b) P.out[b] = deconv_byte<PACKED>(D, P, in0, out0, b);
}

template <bool PACKED>
__global__ __launch_bounds__(256) void k_deconv(deconv_dev D, deconv_plan P) {
  const unsigned long long in0 = P.carry->in, out0 = P.carry->out;
  const unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  const bool r12 = PACKED && deconv_r12_ok(D, P);
  if (4 * g < P.n_bytes) deconv_group4<PACKED>(D, P, in0, out0, g, r12);
  if (g == 0) *P.carry_next = deconv_carry_after<PACKED>(D, P, in0, out0);   // state for the next call
}

// fastlock (dvb.h:428-452 + readerrors, dvb.h:396-417): for every alignment, the number of refills whose bit b
// decodes differently with the alternate inverse polynomial deconv2[b].  readerrors() has its own shift
// register / counters per alignment (in2, n_in2, n_out2); like readbyte its refill schedule is data-independent,
// so refill q of alignment s is a pure function of the input → threads own runs of kErrRun consecutive refills.
constexpr int kErrRun = 32;
struct deconv_err_args {
  unsigned long long deconv[8], deconv2[8];
  unsigned char lut[4][4];
  int pp, pw;
  const lsdr_softsymbol *in;
  unsigned long long refills[4];
  unsigned m0[4];
  deconv_carry *carry[4];        // .in = in2 at call start
  deconv_carry *carry_next[4];
  unsigned long long *errors;    // [4], zeroed by the host
};

__device__ __forceinline__ unsigned long long deconv_err_window(const deconv_err_args &A, int s, unsigned long long in0,
                                                                unsigned long long q) {
  const unsigned long long nsym = A.m0[s] + q * (unsigned)(A.pw / 2);
  unsigned long long w = 0;
  const unsigned take = nsym < 32 ? (unsigned)nsym : 32u;
  for (unsigned k = 0; k < take; ++k) w = (w << 2) | A.lut[s][A.in[nsym - take + k].symbol & 3];
  if (take < 32) w |= in0 << (2 * take);
  return w;
}

__global__ __launch_bounds__(256) void k_deconv_errors(deconv_err_args A) {
  const int s = blockIdx.y;
  const unsigned long long R = A.refills[s];
  const unsigned long long in0 = A.carry[s]->in;
  const unsigned long long q0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * kErrRun;
  unsigned errs = 0;
  if (q0 < R) {
    unsigned long long w = deconv_err_window(A, s, in0, q0);
    for (int r = 0; r < kErrRun && q0 + r < R; ++r) {
      if (r) {
        const unsigned long long nsym = A.m0[s] + (q0 + r) * (unsigned)(A.pw / 2);
        for (int t = A.pw / 2; t > 0; --t) w = (w << 2) | A.lut[s][A.in[nsym - t].symbol & 3];
      }
      for (int b = 0; b < A.pp; ++b) errs += (unsigned)(par64(w & A.deconv2[b]) != par64(w & A.deconv[b]));
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) errs += __shfl_down(errs, d, 64);
  if ((threadIdx.x & 63) == 0 && errs) atomicAdd(A.errors + s, (unsigned long long)errs);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    deconv_carry c;
    c.in = R ? deconv_err_window(A, s, in0, R - 1) : in0;
    c.out = 0;
    *A.carry_next[s] = c;
  }
}

// ======================================================================== mpeg_sync
struct msync_state {    // mpeg_sync members, dvb.h:877-890
  int scan_syncs, want_syncs, fastlock, resync_period;
  unsigned lock_timeout;
  unsigned polarity;
  int resync_phase, bitphase, synchronized, next_sync_count, phase8;
  unsigned lock_timeleft;
  unsigned long long locktime;
};
struct msync_result {
  unsigned long long consumed, produced;
  int n_events, events[4];
  int call_next_sync;
};

__device__ __forceinline__ unsigned char shift_byte(const unsigned char *p, int bitphase) {
  // `w = (w<<8)|*pin; *pout = w >> bitphase` with a 16-bit w (dvb.h:803-807)
  const unsigned short w = (unsigned short)((p[0] << 8) | p[1]);
  return (unsigned char)(w >> bitphase);
}

// search_sync, dvb.h:798-840.  All lanes; returns (uniformly) whether a lock was found.
__device__ bool msync_search(msync_state &S, const unsigned char *in, unsigned long long &pos, unsigned char *tmp,
                             msync_result &R, int *s_best, int tid) {
  const int chunk = kRS * S.scan_syncs;
  for (int k = tid; k < chunk; k += 256) tmp[k] = shift_byte(in + pos + k, S.bitphase);
  __syncthreads();
  if (tid == 0) *s_best = 0x7fffffff;
  __syncthreads();
  int my_pol = 0, my_phase8 = -1;
  if (tid < kRS) {
    int np = 0, nn = 0, p8p = -1, p8n = -1;
    for (int j = 0; j < S.scan_syncs; ++j) {
      const unsigned char b = tmp[tid + j * kRS];
      if (b == kSync) { ++np; p8n = (8 - j) & 7; }
      if (b == kSyncInv) { ++nn; p8p = (8 - j) & 7; }
    }
    int nsyncs;
    if (np > nn) { my_pol = 0; nsyncs = np; my_phase8 = p8p; }
    else { my_pol = 0xff; nsyncs = nn; my_phase8 = p8n; }
    if (nsyncs >= S.want_syncs && my_phase8 >= 0) atomicMin(s_best, tid);   // first offset wins
  }
  __syncthreads();
  const int best = *s_best;
  __syncthreads();
  if (best == 0x7fffffff) {
    // no lock: the reference leaves polarity/phase8 at the values of offset 203 (harmless, kept for state parity)
    if (tid == kRS - 1) { S.polarity = (unsigned)my_pol; S.phase8 = my_phase8; }
    __syncthreads();
    return false;
  }
  if (tid == best) {
    S.polarity = (unsigned)my_pol;
    S.phase8 = my_phase8;
    int i = best;
    if (!i) { i = kRS; S.phase8 = (S.phase8 + 1) & 7; }
    pos += i;
    S.synchronized = 1;
    S.lock_timeleft = S.lock_timeout;
    S.locktime = 0;
    R.events[R.n_events++] = 1;
  }
  __syncthreads();
  return true;
}

// One mpeg_sync::run() call (dvb.h:743-754) by one workgroup of 256: M.S holds the block's state (thread 0 put it there), the call's
// outcome is left in M.R (consumed / produced / events / call_next_sync) and the state in M.S.
struct msync_sh { msync_state S; msync_result R; unsigned long long pos, nout; int s_best, s_stop; };
__device__ void msync_run_body(msync_sh &M, const unsigned char *in, unsigned long long n_in, unsigned char *out, unsigned long long cap, int tid) {
  msync_state &S = M.S;
  msync_result &R = M.R;
  unsigned long long &pos = M.pos, &nout = M.nout;
  int &s_best = M.s_best, &s_stop = M.s_stop;
  __syncthreads();
  if (tid == 0) { R.consumed = R.produced = 0; R.n_events = 0; R.call_next_sync = 0; pos = 0; nout = 0; s_stop = 0; }
  __syncthreads();
  const int chunk = kRS * S.scan_syncs;
  if (S.synchronized) {   // run_decoding, dvb.h:842-875
    // While locked, bit phase and polarity are constant: batches of up to kBatch packets are realigned by all threads,
    // then thread 0 replays the per-packet bookkeeping (sync byte test, lock_timeleft, phase8) over the batch and cuts
    // it where the reference would have dropped the lock.  Same bytes, same events; two barriers per batch, not per packet.
    constexpr unsigned kBatch = 64;
    while (!s_stop) {
      if (n_in - pos < (unsigned long long)kRS + 1 || cap - nout < (unsigned long long)kRS) break;
      unsigned long long P = (n_in - pos - 1) / kRS;
      if (P > (cap - nout) / kRS) P = (cap - nout) / kRS;
      if (P > kBatch) P = kBatch;
      const unsigned char *pin = in + pos;
      unsigned char *pout = out + nout;
      const unsigned nbytes = (unsigned)P * kRS;
      for (unsigned i = tid; i < nbytes; i += 256) pout[i] = (unsigned char)(shift_byte(pin + i, S.bitphase) ^ S.polarity);
      __syncthreads();
      // wave 0: lane p tests packet p's sync byte against the value the 8-packet phase counter predicts for it
      unsigned long long okmask = 0;
      if (tid < 64) {
        bool ok = false;
        if ((unsigned)tid < (unsigned)P) {
          const unsigned char expected = ((S.phase8 + tid) & 7) ? kSync : kSyncInv;
          ok = pout[(unsigned)tid * kRS] == expected;
        }
        okmask = __ballot(ok);
      }
      if (tid == 0) {
        unsigned p = 0;
        for (; p < (unsigned)P; ++p) {
          ++S.locktime;
          if ((okmask >> p) & 1ull) S.lock_timeleft = S.lock_timeout;
          S.phase8 = (S.phase8 + 1) & 7;
          --S.lock_timeleft;
          if (!S.lock_timeleft) {
            S.synchronized = 0;
            S.next_sync_count = 0;
            R.events[R.n_events++] = 0;
            s_stop = 1;
            ++p;
            break;
          }
        }
        pos += (unsigned long long)p * kRS; nout += (unsigned long long)p * kRS;
      }
      __syncthreads();
    }
  } else if (S.fastlock) {   // run_searching_fast, dvb.h:782-796
    bool done = false;
    while (!done && n_in - pos >= (unsigned long long)chunk + 1 && cap - nout >= (unsigned long long)chunk) {
      if (S.resync_phase == 0) {
        for (int bp = 0; bp <= 7 && !done; ++bp) {
          if (tid == 0) S.bitphase = bp;
          __syncthreads();
          done = msync_search(S, in, pos, out + nout, R, &s_best, tid);
        }
        if (!done && tid == 0) S.bitphase = 8;   // loop exit value of the reference's for(bitphase…)
        __syncthreads();
      }
      if (done) break;
      if (tid == 0) { pos += kRS; if (++S.resync_phase >= S.resync_period) S.resync_phase = 0; }
      __syncthreads();
    }
  } else {   // run_searching, dvb.h:756-780
    bool locked_now = false, next_sync = false;
    while (n_in - pos >= (unsigned long long)chunk + 1 && cap - nout >= (unsigned long long)chunk) {
      if (msync_search(S, in, pos, out + nout, R, &s_best, tid)) { locked_now = true; break; }
      if (tid == 0) {
        pos += chunk;
        ++S.bitphase;
        if (S.bitphase == 8) S.bitphase = 0;
      }
      __syncthreads();
      if (S.bitphase == 0) next_sync = true;
      __syncthreads();
    }
    if (!locked_now && next_sync && tid == 0) {
      ++S.next_sync_count;
      if (S.next_sync_count >= 3) { S.next_sync_count = 0; R.call_next_sync = 1; }
    }
  }
  __syncthreads();
  if (tid == 0) { R.consumed = pos; R.produced = nout; }
  __syncthreads();
}
__global__ __launch_bounds__(256) void k_mpeg_sync(msync_state *gS, const unsigned char *in, unsigned long long n_in,
                                                   unsigned char *out, unsigned long long cap, msync_result *gR) {
  __shared__ msync_sh M;
  const int tid = threadIdx.x;
  if (tid == 0) M.S = *gS;
  msync_run_body(M, in, n_in, out, cap, tid);
  if (tid == 0) { *gS = M.S; *gR = M.R; }
}

// Locked path for long inputs (dvb.h:842-875), split so that the byte work runs on the whole chip: while locked, bit phase
// and polarity are constants, so every packet of the run can be realigned independently (k_msync_realign, also records
// whether each packet's sync byte is the one the 8-packet phase counter predicts); k_msync_book then replays the
// per-packet bookkeeping (locktime, lock_timeleft, phase8) over the flags and cuts the run where the reference would
// have dropped the lock.  Bytes realigned past the cut are not part of the output (produced stops there).
__global__ __launch_bounds__(256) void k_msync_realign(const msync_state *gS, const unsigned char *__restrict__ in,
                                                       unsigned long long P, unsigned char *__restrict__ out,
                                                       unsigned char *__restrict__ okflag) {
  const int bitphase = gS->bitphase, phase8 = gS->phase8;
  const unsigned polarity = gS->polarity;
  const unsigned long long nbytes = P * kRS, nthreads = (unsigned long long)gridDim.x * 256;
  const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  for (unsigned long long i = gid; i < nbytes; i += nthreads) out[i] = (unsigned char)(shift_byte(in + i, bitphase) ^ polarity);
  for (unsigned long long p = gid; p < P; p += nthreads) {
    const unsigned char b = (unsigned char)(shift_byte(in + p * kRS, bitphase) ^ polarity);
    const unsigned char expected = ((phase8 + (int)(p & 7)) & 7) ? kSync : kSyncInv;
    okflag[p] = b == expected;
  }
}

__global__ __launch_bounds__(64) void k_msync_book(msync_state *gS, const unsigned char *okflag, unsigned long long P,
                                                   msync_result *gR) {
  msync_state S = *gS;
  msync_result R;
  R.consumed = R.produced = 0; R.n_events = 0; R.call_next_sync = 0;
  for (int k = 0; k < 4; ++k) R.events[k] = 0;
  const int lane = threadIdx.x;
  unsigned long long done = 0;
  bool stop = false;
  for (unsigned long long base = 0; base < P && !stop; base += 64) {
    const unsigned long long left = P - base;
    const unsigned n = left < 64 ? (unsigned)left : 64u;
    const unsigned long long okmask = __ballot((unsigned)lane < n && okflag[base + lane]);
    const unsigned long long full = n == 64 ? ~0ull : ((1ull << n) - 1);
    if (okmask == full && S.lock_timeout > 1) {   // every sync byte as predicted: n packets counted, no lock decision
      S.locktime += n;
      S.phase8 = (S.phase8 + (int)n) & 7;
      S.lock_timeleft = S.lock_timeout - 1;
      done += n;
      continue;
    }
    unsigned p = 0;
    for (; p < n; ++p) {
      ++S.locktime;
      if ((okmask >> p) & 1ull) S.lock_timeleft = S.lock_timeout;
      S.phase8 = (S.phase8 + 1) & 7;
      --S.lock_timeleft;
      if (!S.lock_timeleft) {
        S.synchronized = 0;
        S.next_sync_count = 0;
        R.events[R.n_events++] = 0;
        stop = true;
        ++p;
        break;
      }
    }
    done += p;
  }
  if (lane == 0) {
    R.consumed = R.produced = done * kRS;
    *gS = S; *gR = R;
  }
}

// ======================================================================== deinterleaver
// out[p][j] = in[p*204 + 2244 + j − 12·17·((11 − j) mod 12)]   (dvb.h:935-940)
// A byte of the output comes from up to 2244 bytes further back, so neighbouring 256-byte chunks of the output gather from
// the same input lines: every XCD (block b runs on XCD b mod 8, each with its own L2) walks ONE contiguous eighth of the
// stream, so that an input line is fetched by one L2 instead of by all of them (measured with a round-robin walk: 4 bytes
// fetched per input byte).
__global__ __launch_bounds__(256) void k_deinterleave(const unsigned char *in, unsigned long long n_packets,
                                                      unsigned char *out) {
  const unsigned long long total = n_packets * kRS;
  const unsigned long long chunks = (total + 255) / 256, per_xcd = (chunks + 7) / 8;
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;      // (the grid is a multiple of 8)
  for (unsigned long long c = slot; c < per_xcd; c += slots) {
    const unsigned long long i = (xcd * per_xcd + c) * 256 + threadIdx.x;
    if (i >= total) continue;
    const unsigned long long p = i / kRS;
    const unsigned j = (unsigned)(i % kRS);
    const unsigned delay = 17u * ((11u + 12u * 17u - j) % 12u);
    out[i] = in[p * kRS + 2244 + j - 12ull * delay];
  }
}

// ======================================================================== RS(204,188)
struct gf_tables { unsigned char exp[512]; unsigned char log[256]; };

__device__ __forceinline__ unsigned char gmul(const gf_tables &g, unsigned char x, unsigned char y) {
  return (!x || !y) ? 0 : g.exp[g.log[x] + g.log[y]];
}
__device__ __forceinline__ unsigned char gdiv(const gf_tables &g, unsigned char x, unsigned char y) {
  return !x ? 0 : g.exp[g.log[x] + 255 - g.log[y]];
}
__device__ __forceinline__ unsigned char ginv(const gf_tables &g, unsigned char x) { return g.exp[255 - g.log[x]]; }
__device__ unsigned char eval_poly(const gf_tables &g, const unsigned char *p, int deg, unsigned char x) {
  unsigned char a = 0;
  for (; deg >= 0; --deg) a = gmul(g, a, x) ^ p[deg];
  return a;
}

// Syndromes of one 204-byte packet by one wavefront (rs.h:116-129: synd[j] = P(α^j), P(x) = Σ_i pk[i]·x^(203−i), which the
// reference evaluates by Horner's rule — 204 dependent multiplications per syndrome).  Field arithmetic is exact, so the sum
// may be taken in any order: lane l owns bytes l, l+64, l+128, l+192 and adds pk[i]·α^(j·(203−i)) to all 16 syndromes (one
// log look-up per byte, one exp look-up per term, all independent), then the lanes' partial sums are XORed together.  The
// 16 syndromes come back packed four per word, identical on every lane.
__device__ __forceinline__ void rs_syndromes(const gf_tables &g, const unsigned char *pk, int lane, unsigned s[4]) {
  s[0] = s[1] = s[2] = s[3] = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane + 64 * k;
    const unsigned r = i < kRS ? pk[i] : 0u;
    if (r) {
      const int lr = g.log[r], e = kRS - 1 - i;     // e < 255
      int m = 0;                                    // (j·e) mod 255
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        s[j >> 2] ^= (unsigned)g.exp[lr + m] << (8 * (j & 3));
        m += e;
        m = m >= 255 ? m - 255 : m;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] ^= __shfl_xor(s[q], d, 64);
  }
}

// Key equation of rs_engine::correct (rs.h:173-243) by one lane: Berlekamp-Massey → error locator C (degree L), evaluator
// omega, formal derivative C'.  Results go to LDS for the wavefront's Chien search.
struct rs_key { unsigned char C[16], omega[16], Cprime[16]; int L; };
__device__ void rs_solve_key(const gf_tables &g, const unsigned char *synd, rs_key *K) {
  unsigned char C[16] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned char B[16] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int L = 0, m = 1;
  unsigned char b = 1;
  for (int n = 0; n < 16; ++n) {
    unsigned char d = synd[n];
    for (int i = 1; i <= L; ++i) d ^= gmul(g, C[i], synd[n - i]);
    if (!d) ++m;
    else if (2 * L <= n) {
      unsigned char T[16];
      for (int i = 0; i < 16; ++i) T[i] = C[i];
      for (int i = 0; i < 16 - m; ++i) C[m + i] ^= gmul(g, d, gmul(g, ginv(g, b), B[i]));
      L = n + 1 - L;
      for (int i = 0; i < 16; ++i) B[i] = T[i];
      b = d;
      m = 1;
    } else {
      for (int i = 0; i < 16 - m; ++i) C[m + i] ^= gmul(g, d, gmul(g, ginv(g, b), B[i]));
      ++m;
    }
  }
  unsigned char omega[16];
  for (int i = 0; i < 16; ++i) omega[i] = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j)
      if (i + j < 16) omega[i + j] ^= gmul(g, synd[i], C[j]);
  for (int i = 0; i < 16; ++i) { K->C[i] = C[i]; K->omega[i] = omega[i]; }
  for (int i = 0; i < 15; ++i) K->Cprime[i] = (i & 1) ? 0 : C[i + 1];
  K->Cprime[15] = 0;
  K->L = L;
}

// p(x) for x = α^lx (x ≠ 0), Horner from the highest coefficient like rs.h's eval_poly
__device__ __forceinline__ unsigned char rs_eval(const gf_tables &g, const unsigned char *p, int deg, int lx) {
  unsigned char a = 0;
  for (; deg >= 0; --deg) a = (unsigned char)((a ? g.exp[g.log[a] + lx] : 0) ^ p[deg]);
  return a;
}

// One wavefront per packet, 4 packets per workgroup.  Clean packets (all syndromes zero) cost the parallel syndrome pass only;
// corrupted ones: key equation on lane 0, Chien search + Forney (rs.h:245-264) spread over the lanes (root α^i ↔ lane i mod 64:
// every root touches its own byte, so the corrections commute), then the syndromes of the corrected packet (rs.h:266-269).
// one packet by one wavefront: pk / po / synd / key are THIS wavefront's LDS scratch, the packet is in pk already
__device__ __forceinline__ void rs_decode_wave(const gf_tables &g, unsigned char *pk, unsigned char *po, unsigned char *synd, rs_key *key,
                                               unsigned char *in, unsigned char *out, unsigned long long p, int lane,
                                               unsigned long long *counters /*[0] errs*/, unsigned char *first = nullptr) {
  unsigned s[4];
  rs_syndromes(g, pk, lane, s);
  const bool corrupted = (s[0] | s[1] | s[2] | s[3]) != 0u;      // wave-uniform
  bool still_bad = false;
  if (corrupted) {
    for (int i = lane; i < kTS; i += 64) po[i] = pk[i];   // the message is the first 188 bytes
    if (lane < 16) synd[lane] = (unsigned char)(s[lane >> 2] >> (8 * (lane & 3)));
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) rs_solve_key(g, synd, key);
    __builtin_amdgcn_wave_barrier();
    const rs_key &K = *key;
    // Berlekamp-Massey can end with L = 16 (more errors than the code corrects); C[] and omega[] have 16 coefficients —
    // the reference evaluates degree L regardless and reads one byte past them (rs.h:247,258) — such packets stay
    // uncorrectable either way, so the evaluation is capped at the arrays' degree.
    const int deg = K.L > 15 ? 15 : K.L;
    int nerrs = 0;
    for (int i = lane; i < 255; i += 64) {             // candidate root r = α^i
      if (!rs_eval(g, K.C, deg, i)) {
        const unsigned char xk = g.exp[255 - i];       // 1/r
        const int loc = (255 - i) % 255;
        if (loc < kRS) {
          const unsigned char num = gmul(g, xk, rs_eval(g, K.omega, deg, i));
          const unsigned char den = rs_eval(g, K.Cprime, 14, i);
          const unsigned char e = gdiv(g, num, den);
          nerrs += __popc((unsigned)e);
          if (loc >= 16) po[kRS - 1 - loc] ^= e;
          pk[kRS - 1 - loc] ^= e;
        }
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nerrs += __shfl_xor(nerrs, d, 64);
    if (lane == 0 && nerrs) atomicAdd(&counters[0], (unsigned long long)nerrs);
    __builtin_amdgcn_wave_barrier();
    rs_syndromes(g, pk, lane, s);                  // correct() returns syndromes(pin), rs.h:266-269
    still_bad = (s[0] | s[1] | s[2] | s[3]) != 0u;
    if (lane == 0 && still_bad) po[0] ^= kCorrupt;   // dvb.h:1045
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kTS; i += 64) out[p * kTS + i] = po[i];
    for (int i = lane; i < kRS; i += 64) in[p * kRS + i] = pk[i];   // in-place correction like the reference
    if (first && lane == 0) first[p] = po[0];
  } else {
    for (int i = lane; i < kTS; i += 64) out[p * kTS + i] = pk[i];
  }
}

// One wavefront per packet, 4 packets per workgroup.  Clean packets (all syndromes zero) cost the parallel syndrome pass only;
// corrupted ones: key equation on lane 0, Chien search + Forney (rs.h:245-264) spread over the lanes (root α^i ↔ lane i mod 64:
// every root touches its own byte, so the corrections commute), then the syndromes of the corrected packet (rs.h:266-269).
__global__ __launch_bounds__(256) void k_rs_decode(unsigned char *in, unsigned long long n_packets, unsigned char *out,
                                                   const gf_tables *gtab, unsigned long long *counters /*[0] errs*/) {
  __shared__ gf_tables g;
  __shared__ unsigned char pk[4][kRS + 4], po[4][kTS + 4], synd[4][16];
  __shared__ rs_key key[4];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 512; i += 256) g.exp[i] = gtab->exp[i];
  g.log[tid] = gtab->log[tid];
  const unsigned long long p = (unsigned long long)blockIdx.x * 4 + wv;
  const bool live = p < n_packets;
  if (live)
    for (int i = lane; i < kRS; i += 64) pk[wv][i] = in[p * kRS + i];
  __syncthreads();
  if (!live) return;                                   // (no workgroup barrier below: the rest is per wavefront)
  rs_decode_wave(g, pk[wv], po[wv], synd[wv], &key[wv], in, out, p, lane, counters);
}

// ======================================================================== derandomizer
struct derand_result { unsigned long long produced; int pos_end; };

// Single workgroup: per-packet PRBS offset (resets at 0xB8 / 0xB8^0x55), keep flags, output slots.
__device__ __forceinline__ void derand_scan_body(const unsigned char *in, unsigned n_packets, int pos0,
                                                 const unsigned char *pattern, int *pkt_pos, long long *pkt_dst,
                                                 derand_result *res) {
  constexpr unsigned SEG = 8192, PER = 8;
  __shared__ int s_last[16];       // per wave: index of the last reset (-1: none)
  __shared__ unsigned s_cnt[16];
  __shared__ int carry_last, carry_pos;   // last reset before this segment (absolute index) or -1 with pos0 at index 0
  __shared__ unsigned long long carry_cnt;
  const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) { carry_last = -1; carry_pos = pos0; carry_cnt = 0; }
  for (unsigned base = 0; base < n_packets; base += SEG) {
    __syncthreads();
    // pass 1: reset flags of my 8 packets, running "last reset" (absolute packet index)
    int last = -1;
    unsigned char first[PER];
    for (unsigned q = 0; q < PER; ++q) {
      const unsigned j = base + tid * PER + q;
      unsigned char b0 = 0;
      if (j < n_packets) {
        b0 = in[(unsigned long long)j * kTS];
        if (b0 == kSyncInv || b0 == (kSyncInv ^ kCorrupt)) last = (int)j;
      }
      first[q] = b0;
    }
    // inclusive max-scan of `last` across the workgroup
    int ilast = last;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(ilast, d, 64); if (lane >= (unsigned)d && o > ilast) ilast = o; }
    if (lane == 63) s_last[wv] = ilast;
    __syncthreads();
    int before = carry_last;   // last reset strictly before my first packet
    for (unsigned i = 0; i < wv; ++i) if (s_last[i] > before) before = s_last[i];
    {
      const int up = __shfl_up(ilast, 1, 64);
      if (lane > 0 && up > before) before = up;
    }
    // pass 2: offsets and keep flags
    unsigned keep_cnt = 0;
    int poss[PER];
    bool keeps[PER];
    int cur_last = before;
    for (unsigned q = 0; q < PER; ++q) {
      const unsigned j = base + tid * PER + q;
      keeps[q] = false; poss[q] = 0;
      if (j < n_packets) {
        const unsigned char b0 = first[q];
        if (b0 == kSyncInv || b0 == (kSyncInv ^ kCorrupt)) cur_last = (int)j;
        int pos;
        if (cur_last >= 0) pos = (int)(((long long)(j - (unsigned)cur_last) * kTS) % 1504);
        else pos = (int)(((long long)carry_pos + (long long)j * kTS) % 1504);   // no reset yet: continue from the carried offset
        poss[q] = pos;
        keeps[q] = (unsigned char)(b0 ^ pattern[pos]) == kSync;
        keep_cnt += keeps[q] ? 1u : 0u;
      }
    }
    unsigned icnt = keep_cnt;
    for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(icnt, d, 64); if (lane >= (unsigned)d) icnt += o; }
    if (lane == 63) s_cnt[wv] = icnt;
    __syncthreads();
    unsigned long long off = carry_cnt;
    for (unsigned i = 0; i < wv; ++i) off += s_cnt[i];
    off += icnt - keep_cnt;
    for (unsigned q = 0; q < PER; ++q) {
      const unsigned j = base + tid * PER + q;
      if (j < n_packets) {
        pkt_pos[j] = poss[q];
        pkt_dst[j] = keeps[q] ? (long long)off : -1;
        if (keeps[q]) ++off;
      }
    }
    __syncthreads();
    if (tid == 1023) {
      unsigned long long tot = carry_cnt;
      for (int i = 0; i < 16; ++i) tot += s_cnt[i];
      carry_cnt = tot;
      int l = carry_last;
      for (int i = 0; i < 16; ++i) if (s_last[i] > l) l = s_last[i];
      carry_last = l;
    }
  }
  __syncthreads();
  if (tid == 0) {
    res->produced = carry_cnt;
    // PRBS offset after the last packet
    long long pe;
    if (carry_last >= 0) pe = ((long long)(n_packets - (unsigned)carry_last) * kTS) % 1504;
    else pe = ((long long)pos0 + (long long)n_packets * kTS) % 1504;
    res->pos_end = (int)pe;
  }
}
__global__ __launch_bounds__(1024) void k_derand_scan(const unsigned char *in, unsigned n_packets, int pos0,
                                                      const unsigned char *pattern, int *pkt_pos, long long *pkt_dst,
                                                      derand_result *res) {
  derand_scan_body(in, n_packets, pos0, pattern, pkt_pos, pkt_dst, res);
}

__global__ __launch_bounds__(256) void k_derand_apply(const unsigned char *in, unsigned n_packets, const unsigned char *pattern,
                                                      const int *pkt_pos, const long long *pkt_dst, unsigned char *out) {
  const unsigned p = blockIdx.x * 4 + (threadIdx.x >> 6);
  const unsigned lane = threadIdx.x & 63;
  if (p >= n_packets) return;
  const long long dst = pkt_dst[p];
  if (dst < 0) return;   // restored sync != 0x47: TEI would be set in a slot that is never committed (dvb.h:1149-1156)
  const int pos = pkt_pos[p];
  for (unsigned i = lane; i < (unsigned)kTS; i += 64)
    out[(unsigned long long)dst * kTS + i] = in[(unsigned long long)p * kTS + i] ^ pattern[pos + i];
}

#include "tail_device.h"

}  // namespace

// ============================================================================ host side
namespace {
int log2u(unsigned long long x) { int n = -1; while (x) { ++n; x >>= 1; } return n; }
int hpar(unsigned long long x) { return __builtin_parityll(x); }

struct deconv_host {
  unsigned conv[2], punct[2];
  int pp, pw;
  unsigned long long response[64], deconv[8], deconv2[8];
};
unsigned long long dc_convolve(const deconv_host &d, unsigned long long s) {   // dvb.h:163-179
  const int sbits = log2u(s) + 1;
  unsigned long long iq = 0;
  unsigned char state = 0;
  for (int b = sbits - 1; b >= 0; --b) {
    const unsigned char bit = (s >> b) & 1;
    state = (unsigned char)((state >> 1) | (bit << 6));
    for (int j = 0; j < 2; ++j) {
      const unsigned char xy = (unsigned char)hpar(state & d.conv[j]);
      if (d.punct[j] & (1u << (b % d.pp))) iq = (iq << 1) | xy;
    }
  }
  return iq;
}
void dc_solve(const deconv_host &d, unsigned long long prefix, int nprefix, unsigned long long exp, unsigned long long *best) {
  if (prefix > *best) return;   // dvb.h:205-224
  if (nprefix > 64) return;
  bool solved = true;
  for (int b = 0; b < 64; ++b)
    if (hpar(prefix & d.response[b]) != (int)((exp >> b) & 1)) {
      if (nprefix >= 64 || (d.response[b] >> nprefix) == 0) return;
      solved = false;
    }
  if (solved) { *best = prefix; return; }
  dc_solve(d, prefix, nprefix + 1, exp, best);
  dc_solve(d, prefix | (1ull << nprefix), nprefix + 1, exp, best);
}
}  // namespace

struct lsdr_deconv {
  lsdr_ctx *ctx;
  deconv_host H;
  unsigned char luts[4][4];   // per alignment: I/Q bits by hard symbol
  int locked, skip;
  int n_in[4], n_out[4];      // sync_t counters per alignment (dvb.h:297-306); data-independent → host side
  deconv_carry *d_carry[2];   // [ping-pong][4 alignments]: shift registers live on the device
  int cur[4];
  // fastlock: readerrors() state per alignment (in2 on the device, counters here) and the error totals
  int fastlock;
  unsigned long long deconv2[8];
  int n_in2[4], n_out2[4], cur2[4];
  deconv_carry *d_carry2[2];
  unsigned long long *d_errors;
};

struct lsdr_mpeg_sync {
  lsdr_ctx *ctx;
  msync_state st;             // host mirror (refreshed after every run)
  msync_state *d_state;
  msync_result *d_res;
  unsigned char *d_ok;        // per-packet sync flags of the split locked path
  size_t ok_cap;
  bool report_state;
};

struct lsdr_derandomizer {
  lsdr_ctx *ctx;
  unsigned char pattern[1504];
  unsigned char *d_pattern;
  int pos;                    // PRBS offset carried between calls
  int *d_pkt_pos; long long *d_pkt_dst; size_t cap;
  derand_result *d_res;
};

static void derand_pattern(unsigned char *pattern) {   // dvb.h:1116-1129
  pattern[0] = 0xff;
  unsigned short st = 000251;
  for (int i = 1; i < 188 * 8; ++i) {
    unsigned char o = 0;
    for (int n = 8; n--;) {
      int bit = ((st >> 13) ^ (st >> 14)) & 1;
      o = (unsigned char)((o << 1) | bit);
      st = (unsigned short)((st << 1) | bit);
    }
    pattern[i] = (i % 188) ? o : 0;
  }
}

static void gf_build(gf_tables &g) {   // gf2x_p<u8,u16,0x11d,8,2>, rs.h:47-63
  memset(&g, 0, sizeof(g));            // lut_log[0] is never written by the reference (fresh-heap zero)
  unsigned a = 1;
  for (unsigned i = 0; i < 256; ++i) {
    g.exp[i] = (unsigned char)a;
    g.exp[255 + i] = (unsigned char)a;
    g.log[a] = (unsigned char)i;
    a <<= 1;
    if (a & 256) a ^= 0x11d;
  }
}

static gf_tables *rs_device_tables(lsdr_ctx *c) {   // one copy per context, created on first use, freed with the context
  if (!c->rs_tables) {
    gf_tables g;
    gf_build(g);
    if (hipMalloc(&c->rs_tables, sizeof(gf_tables)) != hipSuccess) { c->rs_tables = nullptr; return nullptr; }
    if (hipMemcpy(c->rs_tables, &g, sizeof(g), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(c->rs_tables);      // never leave a non-null pointer to uninitialised tables behind
      c->rs_tables = nullptr;
      return nullptr;
    }
  }
  return static_cast<gf_tables *>(c->rs_tables);
}

extern "C" {

// ------------------------------------------------------------------ deconvol_sync
int lsdr_deconv_create(lsdr_ctx *c, int rate, int fastlock, lsdr_deconv **out) {
  LSDR_ARG(c && out);
  unsigned pX, pY;
  switch (rate) {   // make_deconvol_sync_simple, dvb.h:480-513
    case LSDR_FEC12: pX = 0x1; pY = 0x1; break;
    case LSDR_FEC23: case LSDR_FEC46: pX = 0xa; pY = 0xf; break;
    case LSDR_FEC34: pX = 0x5; pY = 0x6; break;
    case LSDR_FEC56: pX = 0x15; pY = 0x1a; break;
    case LSDR_FEC78: pX = 0x45; pY = 0x7a; break;
    default: pX = pY = 1;
  }
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_deconv *d = new lsdr_deconv();
  d->ctx = c;
  deconv_host &H = d->H;
  H.conv[0] = 0171; H.conv[1] = 0133;   // DVBS_G1, DVBS_G2 (dvb.h:84-85)
  H.punct[0] = pX; H.punct[1] = pY;
  H.pp = 0; H.pw = 0;
  for (int i = 0; i < 2; ++i) {
    int nbits = log2u(H.punct[i]) + 1;
    if (nbits > H.pp) H.pp = nbits;
    H.pw += __builtin_popcount(H.punct[i]);
  }
  for (int s = 0; s < 64; ++s) H.response[s] = dc_convolve(H, 1ull << s);
  for (int b = 0; b < H.pp; ++b) {
    H.deconv[b] = ~0ull;
    dc_solve(H, 0, 0, (unsigned long long)(1 << b), &H.deconv[b]);
    for (int i = 0; i < 64; ++i)   // D·C = e_b check of the reference (dvb.h:274-283)
      if (hpar(dc_convolve(H, 1ull << i) & H.deconv[b]) != (b == i ? 1 : 0)) {
        delete d;
        lsdr_set_error("deconvol_sync: failed to inverse convolutional coding");
        return LSDR_E_ARG;
      }
  }
  // init_syncs, dvb.h:309-366: lut[re_pos][im_pos]; symbol bit1 ↔ first index, bit0 ↔ second (dvb.h:377)
  for (int id = 0; id < 4; ++id)
    for (int sym = 0; sym < 4; ++sym) {
      const int re_pos = (sym & 2) ? 1 : 0, im_pos = sym & 1, re_neg = !re_pos;
      int I = 0, Q = 0;
      switch (id) {
        case 0: I = re_pos ? 0 : 1; Q = im_pos ? 0 : 1; break;
        case 1: I = im_pos ? 0 : 1; Q = re_neg ? 0 : 1; break;
        case 2: I = re_pos ? 0 : 1; Q = im_pos ? 1 : 0; break;
        case 3: I = im_pos ? 1 : 0; Q = re_neg ? 0 : 1; break;
      }
      d->luts[id][sym] = (unsigned char)((I << 1) | Q);
    }
  d->locked = 0; d->skip = 0;
  d->fastlock = fastlock ? 1 : 0;
  for (int b = 0; b < H.pp; ++b) {   // alternate inverse polynomials of the reference (dvb.h:236-264)
    static const unsigned long long alt[][2] = {
        {0x3baULL, 0x38ccaULL},
        {0xf29ULL, 0x3c569329ULL}, {0x3c552ULL, 0x1dee1cULL}, {0x7948ULL, 0x1e2b49948ULL}, {0x1deULL, 0x1e2a90ULL},
        {0xf247ULL, 0xfd6383bULL}, {0xfd9eeULL, 0xfd91392ULL}, {0xf248d8ULL, 0xfd9eef18ULL},
        {0xf5727fULL, 0x3d5c909758fULL}, {0x3d5c90aaULL, 0x0f5727f0229c90aaULL}, {0x3daa371cULL, 0x3d5f45630ecULL},
        {0xf5727ff48ULL, 0xf57d28260348ULL}, {0xf57d28260ULL, 0xf5727ff48128260ULL},
        {0xfbeac76c454fULL, 0xfb11d6ba045a8fULL}, {0xfb11d6baULL, 0xfbea3c7d930e16baULL},
        {0xfb112d5038dcULL, 0xfb112d5038271cULL}, {0xfbea3c7d68ULL, 0xfbeac7975462a8ULL},
        {0xfb112d50ULL, 0xfbea3c86793290ULL}, {0xfb112dabd2e0ULL, 0xfb112d50c3cd20ULL},
        {0xfb11d640ULL, 0xfbea3c8679c980ULL}};
    d->deconv2[b] = H.deconv[b];
    for (size_t i = 0; i < sizeof(alt) / sizeof(alt[0]); ++i)
      if (H.deconv[b] == alt[i][0]) d->deconv2[b] = alt[i][1];
    if (fastlock && d->deconv2[b] == H.deconv[b]) {
      delete d;
      lsdr_set_error("deconvol_sync: Alt polynomial not provided");   // fail() of dvb.h:265
      return LSDR_E_UNSUPPORTED;
    }
  }
  for (int i = 0; i < 4; ++i) { d->n_in[i] = 0; d->n_out[i] = 0; d->cur[i] = 0; d->n_in2[i] = 0; d->n_out2[i] = 0; d->cur2[i] = 0; }
  for (int i = 0; i < 2; ++i) {
    LSDR_HIP(hipMalloc((void **)&d->d_carry[i], 4 * sizeof(deconv_carry)));
    LSDR_HIP(hipMemset(d->d_carry[i], 0, 4 * sizeof(deconv_carry)));
    LSDR_HIP(hipMalloc((void **)&d->d_carry2[i], 4 * sizeof(deconv_carry)));
    LSDR_HIP(hipMemset(d->d_carry2[i], 0, 4 * sizeof(deconv_carry)));
  }
  LSDR_HIP(hipMalloc((void **)&d->d_errors, 4 * sizeof(unsigned long long)));
  *out = d;
  return LSDR_OK;
}

void lsdr_deconv_destroy(lsdr_deconv *d) {
  if (!d) return;
  (void)hipStreamSynchronize(d->ctx->stream);
  (void)hipFree(d->d_carry[0]); (void)hipFree(d->d_carry[1]);
  (void)hipFree(d->d_carry2[0]); (void)hipFree(d->d_carry2[1]); (void)hipFree(d->d_errors);
  delete d;
}

int lsdr_deconv_reset(lsdr_deconv *d) {   // back to the state after construction (a new stream begins); stream-ordered
  LSDR_ARG(d);
  d->locked = 0; d->skip = 0;
  for (int i = 0; i < 4; ++i) { d->n_in[i] = 0; d->n_out[i] = 0; d->cur[i] = 0; d->n_in2[i] = 0; d->n_out2[i] = 0; d->cur2[i] = 0; }
  for (int i = 0; i < 2; ++i) {
    LSDR_HIP(hipMemsetAsync(d->d_carry[i], 0, 4 * sizeof(deconv_carry), d->ctx->stream));
    LSDR_HIP(hipMemsetAsync(d->d_carry2[i], 0, 4 * sizeof(deconv_carry), d->ctx->stream));
  }
  return LSDR_OK;
}

int lsdr_deconv_next_sync(lsdr_deconv *d) {   // dvb.h:185-193
  LSDR_ARG(d);
  ++d->locked;
  if (d->locked == 4) { d->locked = 0; d->skip = 1; }
  return LSDR_OK;
}

static int deconv_run(lsdr_deconv *d, const lsdr_softsymbol *in, const uint32_t *in_words, size_t in_off, size_t n_in, uint8_t *out,
                      size_t cap_out, size_t *consumed, size_t *produced) {
  LSDR_ARG(d && consumed && produced);
  if (in_words && d->fastlock) { lsdr_set_error("deconvol_sync: fastlock is not available on packed symbols"); return LSDR_E_UNSUPPORTED; }
  const deconv_host &H = d->H;
  size_t pos = (size_t)d->skip;   // in.read(skip), dvb.h:420-421
  d->skip = 0;
  *produced = 0;
  if (pos > n_in) pos = n_in;
  *consumed = pos;
  const size_t readable = n_in - pos;
  if (readable < 64) return LSDR_OK;
  const long long maxrd = (long long)((readable - 64) / (size_t)(H.pw / 2) * (size_t)H.pp / 8);
  long long n = maxrd < (long long)cap_out ? maxrd : (long long)cap_out;
  if (n < 32) return LSDR_OK;   // also covers n == 0
  LSDR_ARG((in || in_words) && out);
  if (d->fastlock) {   // dvb.h:428-452
    deconv_err_args A;
    for (int b = 0; b < 8; ++b) { A.deconv[b] = b < H.pp ? H.deconv[b] : 0; A.deconv2[b] = b < H.pp ? d->deconv2[b] : 0; }
    A.pp = H.pp; A.pw = H.pw;
    A.in = in + pos;
    A.errors = d->d_errors;
    unsigned long long maxR = 0;
    for (int s = 0; s < 4; ++s) {
      for (int k = 0; k < 4; ++k) A.lut[s][k] = d->luts[s][k];
      const long long need2 = 8 * n - d->n_out2[s];
      const unsigned long long R2 = need2 > 0 ? (unsigned long long)((need2 + H.pp - 1) / H.pp) : 0;
      A.refills[s] = R2;
      A.m0[s] = d->n_in2[s] < 64 ? (unsigned)((64 - d->n_in2[s] + 1) / 2) : 0u;
      A.carry[s] = d->d_carry2[d->cur2[s]] + s;
      A.carry_next[s] = d->d_carry2[d->cur2[s] ^ 1] + s;
      if (R2 > maxR) maxR = R2;
    }
    LSDR_HIP(hipMemsetAsync(d->d_errors, 0, 4 * sizeof(unsigned long long), d->ctx->stream));
    const unsigned long long threads = (maxR + kErrRun - 1) / kErrRun;
    hipLaunchKernelGGL(k_deconv_errors, dim3((unsigned)((threads + 255) / 256 ? (threads + 255) / 256 : 1), 4), dim3(256), 0,
                       d->ctx->stream, A);
    LSDR_HIP(hipGetLastError());
    unsigned long long errors[4];
    LSDR_HIP(hipMemcpyAsync(errors, d->d_errors, sizeof(errors), hipMemcpyDeviceToHost, d->ctx->stream));
    LSDR_HIP(hipStreamSynchronize(d->ctx->stream));
    unsigned long errors_best = 1ul << 30;
    int best = 0;
    for (int s = 0; s < 4; ++s) {
      if (A.refills[s]) d->n_in2[s] = 64 - H.pw;
      d->n_out2[s] = (int)(d->n_out2[s] + (long long)A.refills[s] * H.pp - 8 * n);
      d->cur2[s] ^= 1;
      if (errors[s] < errors_best) { errors_best = (unsigned long)errors[s]; best = s; }
    }
    if (best != d->locked) d->locked = best;
    if (errors_best > (unsigned long)(n * 8 / 3)) d->skip = 1;   // deconvolution BER > 33 %: try the next sample alignment
  }
  const int a = d->locked;
  const int n_in0 = d->n_in[a], n_out0 = d->n_out[a];
  long long need = 8 * n - n_out0;
  const unsigned long long R = need > 0 ? (unsigned long long)((need + H.pp - 1) / H.pp) : 0;
  const unsigned m0 = n_in0 < 64 ? (unsigned)((64 - n_in0 + 1) / 2) : 0u;
  const unsigned long long used = R ? m0 + (R - 1) * (unsigned)(H.pw / 2) : 0;

  deconv_dev D;
  for (int b = 0; b < 8; ++b) D.deconv[b] = b < H.pp ? H.deconv[b] : 0;
  for (int sidx = 0; sidx < 4; ++sidx) D.lut[sidx] = d->luts[a][sidx];
  D.pp = H.pp; D.pw = H.pw;
  deconv_plan P;
  P.in = in_words ? nullptr : in + pos;
  P.in_words = in_words; P.in_off = (unsigned long long)(in_off + pos);
  P.out = out;
  P.n_bytes = (unsigned long long)n;
  P.refills = R;
  P.m0 = m0;
  P.n_out0 = n_out0;
  P.carry = d->d_carry[d->cur[a]] + a;
  P.carry_next = d->d_carry[d->cur[a] ^ 1] + a;
  P.n_out_end = (int)(n_out0 + (long long)R * H.pp - 8 * n);
  const unsigned dgrid = (unsigned)(((n + 3) / 4 + 255) / 256);     // a thread owns four consecutive bytes
  if (in_words) hipLaunchKernelGGL(k_deconv<true>, dim3(dgrid), dim3(256), 0, d->ctx->stream, D, P);
  else hipLaunchKernelGGL(k_deconv<false>, dim3(dgrid), dim3(256), 0, d->ctx->stream, D, P);
  LSDR_HIP(hipGetLastError());
  d->cur[a] ^= 1;
  if (R) d->n_in[a] = 64 - H.pw;
  d->n_out[a] = P.n_out_end;
  *consumed = pos + (size_t)used;
  *produced = (size_t)n;
  return LSDR_OK;
}

int lsdr_deconv_run(lsdr_deconv *d, const lsdr_softsymbol *in, size_t n_in, uint8_t *out, size_t cap_out,
                    size_t *consumed, size_t *produced) {
  return deconv_run(d, in, nullptr, 0, n_in, out, cap_out, consumed, produced);
}

int lsdr_deconv_run_hs2(lsdr_deconv *d, const uint32_t *in_words, size_t sym_offset, size_t n_in, uint8_t *out, size_t cap_out,
                        size_t *consumed, size_t *produced) {
  LSDR_ARG(in_words || !n_in);
  return deconv_run(d, nullptr, in_words, sym_offset, n_in, out, cap_out, consumed, produced);
}

// ------------------------------------------------------------------ mpeg_sync
int lsdr_mpeg_sync_create(lsdr_ctx *c, int fastlock, lsdr_mpeg_sync **out) {
  LSDR_ARG(c && out);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_mpeg_sync *m = new lsdr_mpeg_sync();
  m->ctx = c;
  memset(&m->st, 0, sizeof(m->st));
  m->st.scan_syncs = 8; m->st.want_syncs = 4; m->st.lock_timeout = 4;   // dvb.h:727-731
  m->st.fastlock = fastlock ? 1 : 0;
  m->st.resync_period = 1;
  m->report_state = true;
  LSDR_HIP(hipMalloc((void **)&m->d_state, sizeof(msync_state)));
  LSDR_HIP(hipMalloc((void **)&m->d_res, sizeof(msync_result)));
  LSDR_HIP(hipMemcpy(m->d_state, &m->st, sizeof(msync_state), hipMemcpyHostToDevice));
  *out = m;
  return LSDR_OK;
}
void lsdr_mpeg_sync_destroy(lsdr_mpeg_sync *m) {
  if (!m) return;
  (void)hipStreamSynchronize(m->ctx->stream);
  (void)hipFree(m->d_state); (void)hipFree(m->d_res); if (m->d_ok) (void)hipFree(m->d_ok);
  delete m;
}
int lsdr_mpeg_sync_reset(lsdr_mpeg_sync *m) {   // back to the state after construction (options kept); stream-ordered
  LSDR_ARG(m);
  const int fastlock = m->st.fastlock, resync_period = m->st.resync_period;
  memset(&m->st, 0, sizeof(m->st));
  m->st.scan_syncs = 8; m->st.want_syncs = 4; m->st.lock_timeout = 4;   // dvb.h:727-731
  m->st.fastlock = fastlock; m->st.resync_period = resync_period;
  m->report_state = true;
  return lsdr_stage_h2d(m->ctx, m->d_state, &m->st, sizeof(msync_state));
}
int lsdr_mpeg_sync_locked(const lsdr_mpeg_sync *m) { return m ? m->st.synchronized : 0; }
int lsdr_mpeg_sync_set_resync_period(lsdr_mpeg_sync *m, int period) {
  LSDR_ARG(m && period >= 1);
  if (m->st.resync_period == period) return LSDR_OK;
  LSDR_HIP(hipStreamSynchronize(m->ctx->stream));
  LSDR_HIP(hipMemcpy(&m->st, m->d_state, sizeof(msync_state), hipMemcpyDeviceToHost));
  m->st.resync_period = period;
  LSDR_HIP(hipMemcpy(m->d_state, &m->st, sizeof(msync_state), hipMemcpyHostToDevice));
  return LSDR_OK;
}

int lsdr_mpeg_sync_run(lsdr_mpeg_sync *m, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_out,
                       size_t *consumed, size_t *produced, int *events, int *n_events, unsigned long *locktime,
                       int *call_next_sync) {
  LSDR_ARG(m && consumed && produced);
  *consumed = 0; *produced = 0;
  int ne = 0;
  if (m->report_state) {   // "Report unlocked state on first invocation", dvb.h:744-748
    if (events) events[ne] = 0;
    ++ne;
    m->report_state = false;
  }
  if (call_next_sync) *call_next_sync = 0;
  const size_t need_in = m->st.synchronized ? (size_t)kRS + 1 : (size_t)kRS * m->st.scan_syncs + 1;
  const size_t need_out = m->st.synchronized ? (size_t)kRS : (size_t)kRS * m->st.scan_syncs;
  if (n_in >= need_in && cap_out >= need_out) {
    LSDR_ARG(in && out);
    lsdr_ctx *c = m->ctx;
    size_t P = (n_in - 1) / kRS;
    if (P > cap_out / kRS) P = cap_out / kRS;
    if (m->st.synchronized && P >= kMsyncSplitPackets) {
      if (P > m->ok_cap) {
        if (m->d_ok) LSDR_HIP(hipFree(m->d_ok));
        m->d_ok = nullptr; m->ok_cap = 0;
        LSDR_HIP(hipMalloc((void **)&m->d_ok, P + P / 2));
        m->ok_cap = P + P / 2;
      }
      size_t blocks = (P * kRS + 256 * 16 - 1) / (256 * 16);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(k_msync_realign, dim3((unsigned)blocks), dim3(256), 0, c->stream, m->d_state, in,
                         (unsigned long long)P, out, m->d_ok);
      hipLaunchKernelGGL(k_msync_book, dim3(1), dim3(64), 0, c->stream, m->d_state, m->d_ok, (unsigned long long)P, m->d_res);
    } else {
      hipLaunchKernelGGL(k_mpeg_sync, dim3(1), dim3(256), 0, c->stream, m->d_state, in, (unsigned long long)n_in, out,
                         (unsigned long long)cap_out, m->d_res);
    }
    LSDR_HIP(hipGetLastError());
    msync_result r;
    LSDR_HIP(hipMemcpyAsync(&r, m->d_res, sizeof(r), hipMemcpyDeviceToHost, c->stream));
    LSDR_HIP(hipMemcpyAsync(&m->st, m->d_state, sizeof(msync_state), hipMemcpyDeviceToHost, c->stream));
    LSDR_HIP(hipStreamSynchronize(c->stream));
    *consumed = r.consumed;
    *produced = r.produced;
    for (int i = 0; i < r.n_events; ++i) { if (events) events[ne] = r.events[i]; ++ne; }
    if (call_next_sync) *call_next_sync = r.call_next_sync;
  }
  if (n_events) *n_events = ne;
  if (locktime) *locktime = (unsigned long)m->st.locktime;
  return LSDR_OK;
}

// ------------------------------------------------------------------ deinterleaver
int lsdr_deinterleaver_run(lsdr_ctx *c, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_packets,
                           size_t *consumed, size_t *produced) {
  LSDR_ARG(c && consumed && produced);
  *consumed = 0; *produced = 0;
  const size_t window = 17 * 11 * 12 + kRS;   // dvb.h:932
  if (n_in < window) return LSDR_OK;
  size_t n = (n_in - window) / kRS + 1;
  if (n > cap_packets) n = cap_packets;
  if (!n) return LSDR_OK;
  LSDR_ARG(in && out);
  size_t blocks = (n * kRS + 255) / 256;
  const size_t capb = (size_t)c->num_cu * 8;
  if (blocks > capb) blocks = capb;
  blocks = (blocks + 7) / 8 * 8;
  hipLaunchKernelGGL(k_deinterleave, dim3((unsigned)blocks), dim3(256), 0, c->stream, in, (unsigned long long)n, out);
  LSDR_HIP(hipGetLastError());
  *produced = n;
  *consumed = n * kRS;
  return LSDR_OK;
}

// ------------------------------------------------------------------ rs_decoder
int lsdr_rs_decoder_run(lsdr_ctx *c, uint8_t *in, size_t n, uint8_t *out, long *bits, long *errs) {
  LSDR_ARG(c);
  if (bits) *bits = 0;
  if (errs) *errs = 0;
  if (!n) return LSDR_OK;
  LSDR_ARG(in && out);
  LSDR_HIP(hipSetDevice(c->device));
  gf_tables *tab = rs_device_tables(c);
  if (!tab) { lsdr_set_error("rs_decoder: cannot allocate GF tables"); return LSDR_E_NOMEM; }
  if (!c->rs_counter) LSDR_HIP(hipMalloc((void **)&c->rs_counter, sizeof(unsigned long long)));
  LSDR_HIP(hipMemsetAsync(c->rs_counter, 0, sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(k_rs_decode, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, in, (unsigned long long)n, out,
                     (const gf_tables *)tab, c->rs_counter);
  LSDR_HIP(hipGetLastError());
  unsigned long long e = 0;
  LSDR_HIP(hipMemcpyAsync(&e, c->rs_counter, sizeof(e), hipMemcpyDeviceToHost, c->stream));
  LSDR_HIP(hipStreamSynchronize(c->stream));
  if (bits) *bits = (long)(n * kRS * 8);   // nbits += SIZE_RSPACKET*8 per packet, dvb.h:1007
  if (errs) *errs = (long)e;
  return LSDR_OK;
}

void lsdr_rs_tables(uint8_t *exp512, uint8_t *log256, uint8_t *G17) {
  gf_tables g;
  gf_build(g);
  if (exp512) memcpy(exp512, g.exp, 512);
  if (log256) memcpy(log256, g.log, 256);
  if (G17) {   // G(X) = Π (X − α^d), rs.h:93-105
    unsigned char G[17];
    for (int i = 0; i <= 16; ++i) G[i] = (i == 16) ? 1 : 0;
    auto mul = [&](unsigned char x, unsigned char y) -> unsigned char { return (!x || !y) ? 0 : g.exp[g.log[x] + g.log[y]]; };
    for (int dd = 0; dd < 16; ++dd)
      for (int i = 0; i <= 16; ++i) G[i] = (unsigned char)(((i == 16) ? 0 : G[i + 1]) ^ mul(g.exp[dd], G[i]));
    memcpy(G17, G, 17);
  }
}

// ------------------------------------------------------------------ derandomizer
void lsdr_derandomizer_pattern(uint8_t *p) { derand_pattern(p); }

int lsdr_derandomizer_create(lsdr_ctx *c, lsdr_derandomizer **out) {
  LSDR_ARG(c && out);
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_derandomizer *d = new lsdr_derandomizer();
  d->ctx = c;
  derand_pattern(d->pattern);
  d->pos = 0;
  d->d_pkt_pos = nullptr; d->d_pkt_dst = nullptr; d->cap = 0;
  // pattern + 188 bytes of wrap so that pattern[pos + i] never leaves the array
  LSDR_HIP(hipMalloc((void **)&d->d_pattern, 1504 + 188));
  LSDR_HIP(hipMemcpy(d->d_pattern, d->pattern, 1504, hipMemcpyHostToDevice));
  LSDR_HIP(hipMemcpy(d->d_pattern + 1504, d->pattern, 188, hipMemcpyHostToDevice));
  LSDR_HIP(hipMalloc((void **)&d->d_res, sizeof(derand_result)));
  *out = d;
  return LSDR_OK;
}
int lsdr_derandomizer_reset(lsdr_derandomizer *d) {   // a new stream begins
  LSDR_ARG(d);
  d->pos = 0;
  return LSDR_OK;
}
void lsdr_derandomizer_destroy(lsdr_derandomizer *d) {
  if (!d) return;
  (void)hipStreamSynchronize(d->ctx->stream);
  (void)hipFree(d->d_pattern); (void)hipFree(d->d_res); (void)hipFree(d->d_pkt_pos); (void)hipFree(d->d_pkt_dst);
  delete d;
}

int lsdr_derandomizer_run(lsdr_derandomizer *d, const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *consumed,
                          size_t *produced) {
  LSDR_ARG(d && consumed && produced);
  *consumed = 0; *produced = 0;
  if (n > cap) n = cap;   // every input packet needs an output slot (dvb.h:1131)
  if (!n) return LSDR_OK;
  LSDR_ARG(in && out && n < (1ull << 31));
  lsdr_ctx *c = d->ctx;
  LSDR_HIP(hipSetDevice(c->device));
  if (d->cap < n) {
    (void)hipFree(d->d_pkt_pos); (void)hipFree(d->d_pkt_dst);
    LSDR_HIP(hipMalloc((void **)&d->d_pkt_pos, n * sizeof(int)));
    LSDR_HIP(hipMalloc((void **)&d->d_pkt_dst, n * sizeof(long long)));
    d->cap = n;
  }
  hipLaunchKernelGGL(k_derand_scan, dim3(1), dim3(1024), 0, c->stream, in, (unsigned)n, d->pos, (const unsigned char *)d->d_pattern,
                     d->d_pkt_pos, d->d_pkt_dst, d->d_res);
  hipLaunchKernelGGL(k_derand_apply, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, in, (unsigned)n,
                     (const unsigned char *)d->d_pattern, (const int *)d->d_pkt_pos, (const long long *)d->d_pkt_dst, out);
  LSDR_HIP(hipGetLastError());
  derand_result r;
  LSDR_HIP(hipMemcpyAsync(&r, d->d_res, sizeof(r), hipMemcpyDeviceToHost, c->stream));
  LSDR_HIP(hipStreamSynchronize(c->stream));
  d->pos = r.pos_end;
  *consumed = n;
  *produced = (size_t)r.produced;
  return LSDR_OK;
}

}  // extern "C"

#include "tail_host.h"
