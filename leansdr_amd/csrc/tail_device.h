// leansdr_amd/csrc/tail_device.h — the DVB-S FEC tail of a capture batch, resident on the device (included inside fec.hip's anonymous
// namespace): deconvol_sync → mpeg_sync → deinterleaver → rs_decoder → derandomizer (dvb.h:122-513, 712-891, 926-948, 985-1058, 1107-1163)
// for B independent captures, each decoded from freshly constructed blocks, with every data-dependent count — bytes deconvolved, where
// mpeg_sync locked, packets kept — staying in a per-capture record in HBM.  blockIdx.y = capture everywhere; the host reads ONE result
// record per capture when the last kernel has run.  Same kernels' bodies as the one-block-per-call C ABI above (bit-exact blocks):
//
//   k_tail_acquire    one workgroup per capture: the chain's unlocked phase exactly as a scheduler with 64 KiB byte pipes runs it —
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

constexpr unsigned long long kTailWindow = 65536;   // bytes deconvolved per call while mpeg_sync is not locked

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
  const gf_tables *gtab;
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
      if (cap > kTailWindow) cap = kTailWindow;
      unsigned char lut[4];
      s_n = tail_plan_deconv(A, tc, nsym, cap, P, lut, &s_used);
      for (int s = 0; s < 4; ++s) D.lut[s] = lut[s];
      s_in0 = tc.carry[tc.locked].in; s_out0 = tc.carry[tc.locked].out;
    }
    __syncthreads();
    if (!s_n) break;                                                        // the symbols have run out
    for (unsigned long long k = tid; k < s_n; k += 256) P.out[k] = deconv_byte<true>(D, P, s_in0, s_out0, k);
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
  if ((unsigned long long)blockIdx.x * 256 >= P.n_bytes) return;
  deconv_dev D = A.D;
  for (int s = 0; s < 4; ++s) D.lut[s] = tc.plan_lut[s];
  const unsigned long long in0 = tc.plan_in0, out0 = tc.plan_out0;
  for (unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x; k < P.n_bytes; k += (unsigned long long)gridDim.x * 256)
    P.out[k] = deconv_byte<true>(D, P, in0, out0, k);
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
  for (unsigned long long i = gid; i < nbytes; i += nthreads) out[i] = (unsigned char)(shift_byte(in + i, bitphase) ^ polarity);
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
  const unsigned long long chunks = (total + 255) / 256, per_xcd = (chunks + 7) / 8;
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;      // (the grid is a multiple of 8)
  const unsigned char *in = tc.mpeg;
  unsigned char *out = tc.rs;
  for (unsigned long long c = slot; c < per_xcd; c += slots) {
    const unsigned long long i = (xcd * per_xcd + c) * 256 + threadIdx.x;
    if (i >= total) continue;
    const unsigned long long p = i / kRS;
    const unsigned j = (unsigned)(i % kRS);
    const unsigned delay = 17u * ((11u + 12u * 17u - j) % 12u);
    out[i] = in[p * kRS + 2244 + j - 12ull * delay];
  }
}

// rs_decoder<u8,0>::run (dvb.h:998-1053): one wavefront per packet, workgroups walk the capture's packets
__global__ __launch_bounds__(256) void k_tail_rs(tail_args A) {
  tail_cap &tc = A.caps[blockIdx.y];
  const unsigned long long n = tc.n_pk;
  if ((unsigned long long)blockIdx.x * 4 >= n) return;
  __shared__ gf_tables g;
  __shared__ unsigned char pk[4][kRS + 4], po[4][kTS + 4], synd[4][16];
  __shared__ rs_key key[4];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 512; i += 256) g.exp[i] = A.gtab->exp[i];
  g.log[tid] = A.gtab->log[tid];
  __syncthreads();
  for (unsigned long long p = (unsigned long long)blockIdx.x * 4 + wv; p < n; p += (unsigned long long)gridDim.x * 4) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                   // (the wavefront's previous packet is out of its LDS scratch)
    for (int i = lane; i < kRS; i += 64) pk[wv][i] = tc.rs[p * kRS + i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    rs_decode_wave(g, pk[wv], po[wv], synd[wv], &key[wv], tc.rs, tc.rts, p, lane, &tc.rs_errs);
  }
}

// derandomizer::run (dvb.h:1131-1160) from a freshly constructed block: flags, offsets, then the XOR; the capture's result record
__global__ __launch_bounds__(1024) void k_tail_derand_scan(tail_args A) {
  tail_cap &tc = A.caps[blockIdx.y];
  __shared__ derand_result r;
  if (threadIdx.x == 0) { r.produced = 0; r.pos_end = 0; }
  __syncthreads();
  derand_scan_body(tc.rts, (unsigned)tc.n_pk, 0, A.pattern, tc.pkt_pos, tc.pkt_dst, &r);
  __syncthreads();
  if (threadIdx.x == 0) {
    tc.n_ts = r.produced;
    tail_result o;
    o.n_ts = r.produced; o.n_rs = tc.n_pk; o.rs_bit_errors = tc.rs_errs; o.symbols = *tc.nsym;
    o.bytes_deconv = tc.bw; o.bytes_mpeg = tc.mw;
    o.next_sync_calls = tc.next_sync_calls; o.locked_at_end = (unsigned)tc.ms.synchronized; o.alignment = (unsigned)tc.locked;
    o.bitphase = (unsigned)tc.ms.bitphase; o.first_lock_byte = tc.first_lock;
    *tc.res = o;
    __threadfence_system();
  }
}
__global__ __launch_bounds__(256) void k_tail_derand_apply(tail_args A) {
  const tail_cap &tc = A.caps[blockIdx.y];
  const unsigned n = (unsigned)tc.n_pk;
  const unsigned lane = threadIdx.x & 63;
  for (unsigned p = blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += gridDim.x * 4) {
    const long long dst = tc.pkt_dst[p];
    if (dst < 0) continue;   // restored sync != 0x47: TEI would be set in a slot that is never committed (dvb.h:1149-1156)
    const int pos = tc.pkt_pos[p];
    for (unsigned i = lane; i < (unsigned)kTS; i += 64)
      tc.ts[(unsigned long long)dst * kTS + i] = tc.rts[(unsigned long long)p * kTS + i] ^ A.pattern[pos + i];
  }
}

#endif  // LSDR_TAIL_DEVICE_H
