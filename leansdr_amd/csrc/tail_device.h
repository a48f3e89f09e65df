// leansdr_amd/csrc/tail_device.h — the DVB-S FEC tail of a capture batch, resident on the device (included inside fec.hip's anonymous
// namespace): deconvol_sync → mpeg_sync → deinterleaver → rs_decoder → derandomizer (dvb.h:122-513, 712-891, 926-948, 985-1058, 1107-1163)
// for B independent captures, each decoded from freshly constructed blocks, with every data-dependent count — bytes deconvolved, where
// mpeg_sync locked, packets kept — staying in a per-capture record in HBM.  blockIdx.y = capture everywhere; the host reads ONE result
// record per capture when the last kernel has run.  Same kernels' bodies as the one-block-per-call C ABI above (bit-exact blocks):
//
//   k_tail_acquire    one workgroup per capture: the chain's unlocked phase exactly as a scheduler with `window`-byte pipes (8 KiB: the reference's) runs it —
//                     deconvolve a window with the alignment in force, let mpeg_sync search it (and decode what is left of it once it has
//                     locked), next_sync() when mpeg_sync asks for it (dvb.h:185-193, 775-779) — until a window ends locked or the
//                     symbols run out; then the PLAN of the bulk deconvolution (everything that is left, one call)
//   k_tail_deconv     the bulk deconvolution, whole chip
//   k_tail_realign    mpeg_sync's locked path over the bulk (dvb.h:842-875): bit phase and polarity are constants while locked, so
//                     every packet is realigned independently; where the lock would DROP is found in parallel too — the reference's
//                     per-packet countdown drops at the third consecutive sync byte that is not the predicted one after a good one
//   k_tail_book       finishes that call's bookkeeping (one thread), then runs mpeg_sync::run() on what is left until nothing moves
//                     (a dropped lock: search, relock, decode — rare, one workgroup), and fixes the deinterleaver's packet count
//   k_tail_deint / k_tail_rs / k_tail_derand_scan / k_tail_derand_apply   the packet kernels, counts read from the record
#ifndef LSDR_TAIL_DEVICE_H
#define LSDR_TAIL_DEVICE_H


struct tail_result {             // per capture, host-visible when the batch has run
  unsigned long long n_ts;        // TS packets written
  unsigned long long n_rs;        // RS packets decoded
  unsigned long long rs_bit_errors;
  unsigned long long symbols;     // packed decisions the tail was given
  unsigned long long bytes_deconv, bytes_mpeg;
  unsigned next_sync_calls, locked_at_end, alignment, bitphase;
  unsigned long long first_lock_byte;   // deconvolved-stream offset of the first lock (~0: never locked)
};

struct tail_cap {
  const unsigned *words;                 // packed decisions
  const unsigned long long *nsym;        // → how many (device)
  unsigned char *bytes, *mpeg, *rs, *rts, *ts;
  unsigned char *first;                  // [packets] first byte of every RS output packet (what the derandomizer's bookkeeping reads)
  int *pkt_pos; long long *pkt_dst;
  unsigned long long byte_cap, pk_cap;
  tail_result *res;                      // pinned host memory
  // deconvol_sync (dvb.h:297-306): alignment in force, per-alignment counters and shift registers
  int locked, skip;
  int n_in[4], n_out[4];
  deconv_carry carry[4];
  msync_state ms;
  // progress: symbols consumed, bytes deconvolved, bytes mpeg_sync has consumed, bytes it has produced
  unsigned long long pos, bw, br, mw;
  // the bulk call
  deconv_plan plan; unsigned char plan_lut[4]; unsigned long long plan_in0, plan_out0;
  unsigned long long bulk_P;             // packets of mpeg_sync's locked bulk call
  unsigned long long drop_at;            // first packet of the bulk at which the lock drops (≥ bulk_P: never)
  long long last_ok;                     // last packet of the bulk with the predicted sync byte (−1: none)
  unsigned long long n_pk, n_ts, rs_errs, first_lock;
  unsigned next_sync_calls, pad;
};

struct tail_args {
  tail_cap *caps;
  deconv_dev D;                          // polynomials, puncturing (lut filled per capture from `luts`)
  unsigned char luts[4][4];
  msync_state ms0;                       // mpeg_sync as constructed
  unsigned long long window;             // bytes deconvolved per call while mpeg_sync is not locked (the byte pipe of the reference's graph)
  const gf_tables *gtab;
  unsigned char rs_g[16];                // generator polynomial G(x) without its leading 1: coefficients of x^15 … x^0 (rs.h:93-105)
  const unsigned char *pattern;          // derandomizer PRBS (1504 + 188 bytes)
};

// deconvol_sync::run's sizes for one call (dvb.h:419-470), all on the device.  Returns the bytes the call produces (0: nothing moves).
__device__ unsigned long long tail_plan_deconv(const tail_args &A, tail_cap &tc, unsigned long long nsym, unsigned long long cap, deconv_plan &P,
                                               unsigned char *lut, unsigned long long *used_out) {
  const deconv_dev &D = A.D;
  unsigned long long p = (unsigned long long)tc.skip;     // in.read(skip), dvb.h:420-421
  tc.skip = 0;
  const unsigned long long avail = nsym - tc.pos;
  if (p > avail) p = avail;
  tc.pos += p;
  const unsigned long long readable = avail - p;
  *used_out = 0;
  if (readable < 64) return 0;
  const long long maxrd = (long long)((readable - 64) / (unsigned)(D.pw / 2) * (unsigned)D.pp / 8);
  const long long n = maxrd < (long long)cap ? maxrd : (long long)cap;
  if (n < 32) return 0;
  const int a = tc.locked;
  const int n_in0 = tc.n_in[a], n_out0 = tc.n_out[a];
  const long long need = 8 * n - n_out0;
  const unsigned long long R = need > 0 ? (unsigned long long)((need + D.pp - 1) / D.pp) : 0;
  const unsigned m0 = n_in0 < 64 ? (unsigned)((64 - n_in0 + 1) / 2) : 0u;
  P.in = nullptr; P.in_words = tc.words; P.in_off = tc.pos;
  P.out = tc.bytes + tc.bw;
  P.n_bytes = (unsigned long long)n; P.refills = R; P.m0 = m0; P.n_out0 = n_out0;
  P.carry = nullptr; P.carry_next = nullptr;
  P.n_out_end = (int)(n_out0 + (long long)R * D.pp - 8 * n);
  for (int s = 0; s < 4; ++s) lut[s] = A.luts[a][s];
  *used_out = R ? m0 + (R - 1) * (unsigned)(D.pw / 2) : 0;
  return (unsigned long long)n;
}
// … and the counters after it
__device__ void tail_commit_deconv(const tail_args &A, tail_cap &tc, const deconv_plan &P, unsigned long long used) {
  const int a = tc.locked;
  if (P.refills) tc.n_in[a] = 64 - A.D.pw;
  tc.n_out[a] = P.n_out_end;
  tc.pos += used;
  tc.bw += P.n_bytes;
}

__global__ __launch_bounds__(256) void k_tail_acquire(tail_args A) {
  tail_cap &tc = A.caps[blockIdx.y];
  __shared__ msync_sh M;
  __shared__ deconv_plan P;
  __shared__ deconv_dev D;
  __shared__ unsigned long long s_n, s_used, s_in0, s_out0;
  __shared__ int s_again;
  const int tid = threadIdx.x;
  const unsigned long long nsym = *tc.nsym;
  if (tid == 0) {
    tc.locked = 0; tc.skip = 0;
    for (int i = 0; i < 4; ++i) { tc.n_in[i] = 0; tc.n_out[i] = 0; tc.carry[i].in = 0; tc.carry[i].out = 0; }
    tc.pos = tc.bw = tc.br = tc.mw = 0;
    tc.n_pk = tc.n_ts = tc.rs_errs = 0; tc.next_sync_calls = 0; tc.first_lock = ~0ull;
    tc.bulk_P = 0; tc.drop_at = ~0ull; tc.last_ok = -1;
    M.S = A.ms0;
    D = A.D;
  }
  __syncthreads();
  while (true) {
    if (M.S.synchronized) break;                                            // (uniform: shared)
    if (tid == 0) {
      unsigned long long cap = tc.byte_cap - tc.bw;
      if (cap > A.window) cap = A.window;
      unsigned char lut[4];
      s_n = tail_plan_deconv(A, tc, nsym, cap, P, lut, &s_used);
      for (int s = 0; s < 4; ++s) D.lut[s] = lut[s];
      s_in0 = tc.carry[tc.locked].in; s_out0 = tc.carry[tc.locked].out;
    }
    __syncthreads();
    if (!s_n) break;                                                        // the symbols have run out
    {
      const bool r12 = deconv_r12_ok(D, P);
      for (unsigned long long g = tid; 4 * g < s_n; g += 256) deconv_group4<true>(D, P, s_in0, s_out0, g, r12);
    }
    __syncthreads();
    if (tid == 0) {
      tc.carry[tc.locked] = deconv_carry_after<true>(D, P, s_in0, s_out0);
      tail_commit_deconv(A, tc, P, s_used);
    }
    __threadfence_block();
    __syncthreads();
    // mpeg_sync::run() until nothing moves (the scheduler's fixpoint over this window)
    do {
      const bool was = M.S.synchronized != 0;
      __syncthreads();
      msync_run_body(M, tc.bytes + tc.br, tc.bw - tc.br, tc.mpeg + tc.mw, tc.byte_cap - tc.mw, tid);
      if (tid == 0) {
        if (!was && M.S.synchronized && tc.first_lock == ~0ull) tc.first_lock = tc.br + M.R.consumed;
        if (M.R.call_next_sync) {                                           // deconvol_sync::next_sync, dvb.h:185-193
          ++tc.next_sync_calls;
          ++tc.locked;
          if (tc.locked == 4) { tc.locked = 0; tc.skip = 1; }
        }
        tc.br += M.R.consumed; tc.mw += M.R.produced;
        s_again = (M.R.consumed || M.R.produced) ? 1 : 0;
      }
      __syncthreads();
    } while (s_again);
  }
  __syncthreads();
  // the bulk call: everything that is left, with the alignment in force (only reached locked, or with nothing left)
  if (tid == 0) {
    unsigned long long used = 0;
    deconv_plan BP;
    unsigned char lut[4];
    BP.n_bytes = 0; BP.refills = 0;
    unsigned long long n = 0;
    if (M.S.synchronized) n = tail_plan_deconv(A, tc, nsym, tc.byte_cap - tc.bw, BP, lut, &used);
    if (n) {
      tc.plan = BP;
      for (int s = 0; s < 4; ++s) tc.plan_lut[s] = lut[s];
      tc.plan_in0 = tc.carry[tc.locked].in; tc.plan_out0 = tc.carry[tc.locked].out;
      tail_commit_deconv(A, tc, BP, used);
    } else {
      tc.plan.n_bytes = 0;
    }
    // mpeg_sync's locked call over [br, bw): whole packets with one byte of look-ahead, room permitting (dvb.h:842-846)
    unsigned long long Pk = 0;
    if (M.S.synchronized && tc.bw - tc.br >= (unsigned long long)kRS + 1) {
      Pk = (tc.bw - tc.br - 1) / kRS;
      const unsigned long long room = (tc.byte_cap - tc.mw) / kRS;
      if (Pk > room) Pk = room;
    }
    tc.bulk_P = Pk;
    tc.ms = M.S;
  }
}

__global__ __launch_bounds__(256) void k_tail_deconv(tail_args A) {
  const tail_cap &tc = A.caps[blockIdx.y];
  const deconv_plan P = tc.plan;
  if ((unsigned long long)blockIdx.x * 1024 >= P.n_bytes) return;
  deconv_dev D = A.D;
  for (int s = 0; s < 4; ++s) D.lut[s] = tc.plan_lut[s];
  const unsigned long long in0 = tc.plan_in0, out0 = tc.plan_out0;
  const bool r12 = deconv_r12_ok(D, P);
  // four groups of four bytes per trip, their sixteen packed words requested together (a thread's trips are dependent round trips to
  // memory otherwise: 0.5 ms for 16 captures at 32 waves per CU)
  constexpr int U = 4;
  const unsigned long long stride = (unsigned long long)gridDim.x * 256;
  const unsigned *__restrict__ words = P.in_words;
  for (unsigned long long g = (unsigned long long)blockIdx.x * 256 + threadIdx.x; 4 * g < P.n_bytes; g += U * stride) {
    unsigned W[U][4];
    unsigned long long wi[U]; int sh[U]; bool fast[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long gu = g + u * stride;
      fast[u] = r12 && 4 * gu < P.n_bytes && deconv_r12_addr(P, 4 * gu, wi[u], sh[u]);
      if (fast[u]) { W[u][0] = words[wi[u] - 3]; W[u][1] = words[wi[u] - 2]; W[u][2] = words[wi[u] - 1]; W[u][3] = words[wi[u]]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long gu = g + u * stride;
      if (fast[u]) deconv_store4(P.out + 4 * gu, deconv_r12_bits(D, W[u][0], W[u][1], W[u][2], W[u][3], sh[u]));
      else if (4 * gu < P.n_bytes) deconv_group4<true>(D, P, in0, out0, gu, false);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) A.caps[blockIdx.y].carry[tc.locked] = deconv_carry_after<true>(D, P, in0, out0);
}

// mpeg_sync locked over the bulk.  Packet p's sync byte is "ok" when it is the one the 8-packet phase counter predicts (dvb.h:853-862);
// lock_timeleft is set to lock_timeout by an ok packet and counted down by every packet: with q the last ok packet at or before p
// it is lock_timeout − 1 − (p − q) after packet p — the lock drops at the first p with p − q = lock_timeout − 1, or, before any ok
// packet, when the countdown the call started with runs out.
__global__ __launch_bounds__(256) void k_tail_realign(tail_args A) {
  tail_cap &tc = A.caps[blockIdx.y];
  const unsigned long long P = tc.bulk_P;
  if ((unsigned long long)blockIdx.x * 256 >= P * kRS) return;
  const int bitphase = tc.ms.bitphase, phase8 = tc.ms.phase8;
  const unsigned polarity = tc.ms.polarity;
  const unsigned tmo = tc.ms.lock_timeout, tl0 = tc.ms.lock_timeleft;
  const unsigned char *in = tc.bytes + tc.br;
  unsigned char *out = tc.mpeg + tc.mw;
  const unsigned long long nbytes = P * kRS, nthreads = (unsigned long long)gridDim.x * 256;
  const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  // four output bytes per trip from five input bytes (two or three aligned dwords), one dword store (the output of a locked run starts
  // on a packet boundary of the 4-byte-aligned mpeg buffer)
  if ((((unsigned long long)out) & 3ull) == 0) {
    const unsigned pol4 = polarity * 0x01010101u;
    for (unsigned long long g = gid; 4 * g < nbytes; g += nthreads) {
      const unsigned long long a = (unsigned long long)(in + 4 * g);
      const unsigned *ap = reinterpret_cast<const unsigned *>(a & ~3ull);
      const int sh = (int)(a & 3ull) * 8;
      const unsigned d0 = ap[0], d1 = ap[1], d2 = sh ? ap[2] : 0u;        // (bytes + byte_cap + 64: room behind the last byte)
      const unsigned lo = __builtin_amdgcn_alignbit(d1, d0, sh), hi = __builtin_amdgcn_alignbit(d2, d1, sh);
      unsigned r = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned b0 = (lo >> (8 * j)) & 255u, b1 = j < 3 ? (lo >> (8 * j + 8)) & 255u : hi & 255u;
        r |= ((((b0 << 8) | b1) >> bitphase) & 255u) << (8 * j);
      }
      reinterpret_cast<unsigned *>(out)[g] = r ^ pol4;
    }
  } else {
    for (unsigned long long i = gid; i < nbytes; i += nthreads) out[i] = (unsigned char)(shift_byte(in + i, bitphase) ^ polarity);
  }
  auto ok = [&](unsigned long long p) {
    const unsigned char b = (unsigned char)(shift_byte(in + p * kRS, bitphase) ^ polarity);
    return b == (((phase8 + (int)(p & 7)) & 7) ? kSync : kSyncInv);
  };
  unsigned long long drop = ~0ull;
  long long last = -1;
  for (unsigned long long p = gid; p < P; p += nthreads) {
    if (ok(p)) { last = (long long)p; continue; }
    // p is a miss: does the countdown reach zero here?
    bool d = false;
    if (p + 1 >= tmo - 1 + 1 && tmo >= 2) {              // an ok packet tmo−1 back, misses since
      bool all_miss = true;
      for (unsigned t = 1; t + 1 < tmo && all_miss; ++t) all_miss = !ok(p - t);
      d = all_miss && p >= tmo - 1 && ok(p - (tmo - 1));
    }
    if (!d && p + 1 == tl0) {                             // no ok packet since the call began and the carried countdown ends here
      bool all_miss = true;
      for (unsigned long long t = 0; t < p && all_miss; ++t) all_miss = !ok(t);
      d = all_miss;
    }
    if (d && p < drop) drop = p;
  }
  if (tmo < 2) {                                          // (lock_timeout 1: every packet ends the lock; not a configuration leandvb uses)
    if (gid == 0 && P) drop = 0;
  }
#pragma unroll
  for (int dd = 32; dd >= 1; dd >>= 1) {
    const unsigned long long od = __shfl_xor(drop, dd, 64);
    const long long ol = __shfl_xor(last, dd, 64);
    if (od < drop) drop = od;
    if (ol > last) last = ol;
  }
  if ((threadIdx.x & 63) == 0) {
    if (drop != ~0ull) atomicMin(&tc.drop_at, drop);
    if (last >= 0) atomicMax(&tc.last_ok, last);
  }
}

__global__ __launch_bounds__(256) void k_tail_book(tail_args A) {
  tail_cap &tc = A.caps[blockIdx.y];
  __shared__ msync_sh M;
  __shared__ int s_again;
  const int tid = threadIdx.x;
  if (tid == 0) {
    msync_state S = tc.ms;
    const unsigned long long P = tc.bulk_P;
    if (P) {
      unsigned long long done = P;
      if (tc.drop_at < P) {                               // the lock drops at packet drop_at (dvb.h:866-871)
        done = tc.drop_at + 1;
        S.synchronized = 0; S.next_sync_count = 0; S.lock_timeleft = 0;
      } else if (tc.last_ok >= 0) {
        S.lock_timeleft = S.lock_timeout - 1 - (unsigned)(P - 1 - (unsigned long long)tc.last_ok);
      } else {
        S.lock_timeleft -= (unsigned)P;
      }
      S.locktime += done;
      S.phase8 = (int)((S.phase8 + done) & 7);
      tc.br += done * kRS; tc.mw += done * kRS;
    }
    M.S = S;
  }
  __syncthreads();
  // whatever is left: the end of the stream, or a dropped lock — mpeg_sync::run() until nothing moves.  (A next_sync() asked for here
  // changes nothing any more: every symbol has been deconvolved.)
  do {
    __syncthreads();
    msync_run_body(M, tc.bytes + tc.br, tc.bw - tc.br, tc.mpeg + tc.mw, tc.byte_cap - tc.mw, tid);
    if (tid == 0) {
      if (M.R.call_next_sync) { ++tc.next_sync_calls; ++tc.locked; if (tc.locked == 4) { tc.locked = 0; tc.skip = 1; } }
      tc.br += M.R.consumed; tc.mw += M.R.produced;
      s_again = (M.R.consumed || M.R.produced) ? 1 : 0;
    }
    __syncthreads();
  } while (s_again);
  if (tid == 0) {
    tc.ms = M.S;
    // deinterleaver<u8>::run (dvb.h:926-948): packets while 17·11·12 + 204 bytes are readable
    const unsigned long long window = 17 * 11 * 12 + kRS;
    unsigned long long n = tc.mw >= window ? (tc.mw - window) / kRS + 1 : 0;
    if (n > tc.pk_cap) n = tc.pk_cap;
    tc.n_pk = n;
  }
}

__global__ __launch_bounds__(256) void k_tail_deint(tail_args A) {
  const tail_cap &tc = A.caps[blockIdx.y];
  const unsigned long long total = tc.n_pk * kRS;
  const unsigned long long chunks = (total + 1023) / 1024, per_xcd = (chunks + 7) / 8;
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;      // (the grid is a multiple of 8)
  const unsigned char *in = tc.mpeg;
  unsigned char *out = tc.rs;
  // four consecutive bytes of a packet per thread (a packet is 51 dwords): four gathered bytes, one dword store
  for (unsigned long long c = slot; c < per_xcd; c += slots) {
    const unsigned long long i = ((xcd * per_xcd + c) * 256 + threadIdx.x) * 4;
    if (i >= total) continue;
    const unsigned long long p = i / kRS;
    const unsigned j = (unsigned)(i % kRS);
    unsigned r = 0;
#pragma unroll
    for (unsigned u = 0; u < 4; ++u) {
      const unsigned delay = 17u * ((11u + 12u * 17u - (j + u)) % 12u);
      r |= (unsigned)in[p * kRS + 2244 + j + u - 12ull * delay] << (8 * u);
    }
    reinterpret_cast<unsigned *>(out)[i >> 2] = r;
  }
}

// rs_decoder<u8,0>::run (dvb.h:998-1053).  A packet is a codeword — all 16 syndromes zero (rs.h:116-129) — exactly when its polynomial
// is divisible by the generator G(x) = Π (x − α^j) (rs.h:93-105), i.e. when the remainder of the systematic encoder's division is zero.
// That division is a byte-wise LFSR: R ← (R·x^8) ⊕ T[b ⊕ top(R)], T[f] = f·(G − x^16), ONE 16-byte table row per byte where the
// syndromes cost 16 log/exp look-ups.  So: a wavefront stages 64 consecutive packets in LDS (coalesced 16-byte loads), every LANE
// divides its own packet (its bytes sit 204 apart: 51 dwords, odd — conflict-free), the 64 messages leave as one contiguous stretch,
// and only packets with a non-zero remainder (channel errors) go through the syndrome / Berlekamp-Massey / Chien path of
// rs_decode_wave, one at a time by the whole wavefront.  Same bytes and counters as k_rs_decode.
constexpr int kRsChunk = 64;                              // packets per wavefront pass
__global__ __launch_bounds__(256) void k_tail_rs(tail_args A) {
  tail_cap &tc = A.caps[blockIdx.y];
  const unsigned long long n = tc.n_pk;
  if ((unsigned long long)blockIdx.x * 4 * kRsChunk >= n) return;
  __shared__ gf_tables g;
  __shared__ __attribute__((aligned(16))) uint4 T[4][256];          // T[k][f] = f·x^(16+k) mod G: four packet bytes per division step
  __shared__ __attribute__((aligned(16))) unsigned char stage[4][kRsChunk * kRS];
  __shared__ unsigned char po[4][kTS + 4], synd[4][16];
  __shared__ rs_key key[4];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 512; i += 256) g.exp[i] = A.gtab->exp[i];
  g.log[tid] = A.gtab->log[tid];
  __syncthreads();
  {
    // row f of the division table, laid out like R: R[m] (coefficient of x^(15−m)) is byte 15 − m of the 128-bit little-endian value
    unsigned char row[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) row[15 - m] = gmul(g, (unsigned char)tid, A.rs_g[m]);
    uint4 t;
    __builtin_memcpy(&t, row, 16);
    T[0][tid] = t;
  }
  __syncthreads();
  for (int k = 1; k < 4; ++k) {                          // T[k][f] = T[k−1][f]·x mod G: one more step of the byte-wise division
    const uint4 r = T[k - 1][tid], t0 = T[0][r.w >> 24];
    uint4 n;
    n.w = __builtin_amdgcn_alignbyte(r.w, r.z, 3) ^ t0.w; n.z = __builtin_amdgcn_alignbyte(r.z, r.y, 3) ^ t0.z;
    n.y = __builtin_amdgcn_alignbyte(r.y, r.x, 3) ^ t0.y; n.x = (r.x << 8) ^ t0.x;
    T[k][tid] = n;
    __syncthreads();
  }
  unsigned char *const st = stage[wv];
  for (unsigned long long c = (unsigned long long)blockIdx.x * 4 + wv; c * kRsChunk < n; c += (unsigned long long)gridDim.x * 4) {
    const unsigned long long p0 = c * kRsChunk;
    const unsigned np = n - p0 < (unsigned long long)kRsChunk ? (unsigned)(n - p0) : (unsigned)kRsChunk;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                   // (the wavefront's previous pass is out of its LDS)
    const uint4 *src = reinterpret_cast<const uint4 *>(tc.rs + p0 * kRS);       // 64·204 bytes = 816 16-byte pieces; rs is 256-byte aligned
    const unsigned pieces = (np * kRS + 15) / 16;      // (the buffer has room for whole chunks: pk_cap is padded)
    for (unsigned i = lane; i < pieces; i += 64) reinterpret_cast<uint4 *>(st)[i] = src[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    if ((unsigned)lane < np) {
      // four bytes per step: the register's top four coefficients ⊕ the next four packet bytes leave through four INDEPENDENT table
      // rows (R·x^4 + b0·x^19 + … + b3·x^16 mod G), 51 dependent steps per packet instead of 204
      const unsigned *pk = reinterpret_cast<const unsigned *>(st + lane * kRS);
#pragma unroll 3
      for (int i = 0; i < kRS / 4; ++i) {
        const unsigned t = __builtin_bswap32(pk[i]) ^ r3;
        const uint4 a = T[3][t >> 24], b = T[2][(t >> 16) & 255u], c = T[1][(t >> 8) & 255u], d = T[0][t & 255u];
        r3 = r2 ^ a.w ^ b.w ^ c.w ^ d.w;
        r2 = r1 ^ a.z ^ b.z ^ c.z ^ d.z;
        r1 = r0 ^ a.y ^ b.y ^ c.y ^ d.y;
        r0 = a.x ^ b.x ^ c.x ^ d.x;
      }
    }
    const unsigned long long bad = __ballot((r0 | r1 | r2 | r3) != 0u);
    // the 64 messages: 188 = 47 dwords of every 51-dword packet, contiguous in the output
    unsigned *dst = reinterpret_cast<unsigned *>(tc.rts + p0 * kTS);
    const unsigned words = np * (kTS / 4);
    for (unsigned o = lane; o < words; o += 64) {
      const unsigned pkt = o / 47u, j = o - pkt * 47u;
      const unsigned w = reinterpret_cast<const unsigned *>(st)[pkt * 51u + j];
      dst[o] = w;
      if (j == 0) tc.first[p0 + pkt] = (unsigned char)w;
    }
    for (unsigned long long m = bad; m; m &= m - 1) {  // wave-uniform: the packets the channel damaged, by the whole wavefront
      const int q = __builtin_ctzll(m);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      rs_decode_wave(g, st + q * kRS, po[wv], synd[wv], &key[wv], tc.rs, tc.rts, p0 + q, lane, &tc.rs_errs, tc.first);
    }
  }
}

// derandomizer::run (dvb.h:1131-1160) from a freshly constructed block.  The PRBS offset of packet p is 188·(p − r) mod 1504 with r the last
// packet at or before p whose first byte is the inverted sync (0xB8, or 0xB8 ^ 0x55 where rs_decoder marked it uncorrectable) — 0 + 188·p
// before the first one; a packet is kept when its restored first byte is 0x47.  One workgroup per capture, ONE pass: every thread owns a
// run of consecutive packets (their first bytes were set aside by k_tail_rs: contiguous), the runs' "last reset" and "kept" totals are
// scanned across the workgroup, then every thread walks its run again.  The capture's result record is written here.
__global__ __launch_bounds__(1024) void k_tail_derand_scan(tail_args A) {
  tail_cap &tc = A.caps[blockIdx.y];
  __shared__ long long s_last[16];
  __shared__ unsigned s_cnt[16];
  const unsigned n = (unsigned)tc.n_pk, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned per = (n + 1023u) / 1024u, lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
  const unsigned char *first = tc.first;
  auto is_reset = [](unsigned char b) { return b == kSyncInv || b == (kSyncInv ^ kCorrupt); };
  // pass 1: the last reset inside my run
  long long last = -1;
  for (unsigned p = lo; p < hi; ++p) if (is_reset(first[p])) last = (long long)p;
  long long ilast = last;                                   // inclusive max-scan
  for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(ilast, d, 64); if (lane >= (unsigned)d && o > ilast) ilast = o; }
  if (lane == 63) s_last[wv] = ilast;
  __syncthreads();
  long long before = -1;                                    // last reset strictly in front of my run
  for (unsigned i = 0; i < wv; ++i) if (s_last[i] > before) before = s_last[i];
  { const long long up = __shfl_up(ilast, 1, 64); if (lane > 0 && up > before) before = up; }
  // pass 2a: how many of my packets are kept
  auto pos_of = [](unsigned p, long long r) { return (int)((r >= 0 ? (long long)(p - (unsigned)r) * kTS : (long long)p * kTS) % 1504); };
  unsigned keep = 0;
  {
    long long r = before;
    for (unsigned p = lo; p < hi; ++p) {
      const unsigned char b0 = first[p];
      if (is_reset(b0)) r = (long long)p;
      keep += (unsigned char)(b0 ^ A.pattern[pos_of(p, r)]) == kSync ? 1u : 0u;
    }
  }
  unsigned icnt = keep;
  for (int d = 1; d < 64; d <<= 1) { const unsigned o = __shfl_up(icnt, d, 64); if (lane >= (unsigned)d) icnt += o; }
  if (lane == 63) s_cnt[wv] = icnt;
  __syncthreads();
  unsigned long long off = 0;
  for (unsigned i = 0; i < wv; ++i) off += s_cnt[i];
  off += icnt - keep;
  // pass 2b: offsets and destinations
  {
    long long r = before;
    for (unsigned p = lo; p < hi; ++p) {
      const unsigned char b0 = first[p];
      if (is_reset(b0)) r = (long long)p;
      const int pos = pos_of(p, r);
      const bool k = (unsigned char)(b0 ^ A.pattern[pos]) == kSync;
      tc.pkt_pos[p] = pos;
      tc.pkt_dst[p] = k ? (long long)off : -1;
      off += k ? 1u : 0u;
    }
  }
  if (tid == 1023) {
    unsigned long long tot = 0;
    for (int i = 0; i < 16; ++i) tot += s_cnt[i];
    tc.n_ts = tot;
    tail_result o;
    o.n_ts = tot; o.n_rs = tc.n_pk; o.rs_bit_errors = tc.rs_errs; o.symbols = *tc.nsym;
    o.bytes_deconv = tc.bw; o.bytes_mpeg = tc.mw;
    o.next_sync_calls = tc.next_sync_calls; o.locked_at_end = (unsigned)tc.ms.synchronized; o.alignment = (unsigned)tc.locked;
    o.bitphase = (unsigned)tc.ms.bitphase; o.first_lock_byte = tc.first_lock;
    *tc.res = o;
    __threadfence_system();
  }
}
// the XOR, 47 dwords per packet (packets, pattern offsets and destinations are all multiples of 4 bytes)
__global__ __launch_bounds__(256) void k_tail_derand_apply(tail_args A) {
  const tail_cap &tc = A.caps[blockIdx.y];
  const unsigned n = (unsigned)tc.n_pk;
  const unsigned lane = threadIdx.x & 63;
  const unsigned *in = reinterpret_cast<const unsigned *>(tc.rts);
  unsigned *out = reinterpret_cast<unsigned *>(tc.ts);
  for (unsigned p = blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += gridDim.x * 4) {
    const long long dst = tc.pkt_dst[p];
    if (dst < 0) continue;   // restored sync != 0x47: TEI would be set in a slot that is never committed (dvb.h:1149-1156)
    const int pos = tc.pkt_pos[p];
    if (lane < 47) out[(unsigned long long)dst * 47 + lane] = in[(unsigned long long)p * 47 + lane] ^ reinterpret_cast<const unsigned *>(A.pattern + pos)[lane];
  }
}

#endif  // LSDR_TAIL_DEVICE_H
