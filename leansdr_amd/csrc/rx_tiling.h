// leansdr_amd/csrc/rx_tiling.h — device-side seam reconciliation shared by the time-tiled (throughput-mode)
// receivers: cstln_receiver (cstln_receiver.hip) and fast_qpsk_receiver (hs.hip).  Included inside each file's
// anonymous namespace.  SYM = the symbol record type, STATE = the carried loop state (its phase is rotated back into
// tile 0's frame at the end of a run through rx_rotate_back(), specialised by each includer, as is rx_freq_tap():
// the receiver's freq_tap in cycles per sample, which rides home with the run's totals).
#ifndef LSDR_RX_TILING_H
#define LSDR_RX_TILING_H

template <typename SYM>
struct rx_tile_info_t {       // per tile, written by the tile kernel
  float mu_begin, phase_begin;   // state at the start of the tile body (after warm-up)
  float mu_end, phase_end;       // state at the end of the tile body
  unsigned count;                // symbols emitted in the body
  SYM pre;                       // last symbol of the warm-up (needed when a seam loses a symbol)
  unsigned has_pre;
  unsigned n_warm;               // symbols of the LAST warm-up chunk, kept in the warm-up staging row of the tile
};

// Decision overlap of a seam: tile j's last warm-up chunk and tile j−1's body cover the same samples, so their symbol
// decisions agree up to the quadrant step k between the two carrier frames.  k by majority over the last kSeamVote
// symbol pairs (relabel[k][cur] == prev) is far more robust than rounding the difference of two noisy phase registers.
constexpr int kSeamVote = 8;

struct rx_tile_fix {          // per tile, produced by the seam pass
  unsigned long long out_offset; // where the tile's (fixed-up) symbols start in the output (block-local)
  unsigned rot;                  // quadrant correction (block-local)
  unsigned drop_first;           // 1: first body symbol duplicates the previous tile's last
  unsigned insert_pre;           // 1: the warm-up's last symbol belongs to this tile
};

// Seam pass: reconciles neighbouring tiles on the device.
//  * carrier quadrant: tile j locked k_j·(65536/R) away from where tile j−1 ended → running
//    rotation (prefix sum mod R) used to relabel its symbols;
//  * symbol timing: mu at the start of tile j vs mu at the end of tile j−1 differ by ≈ ±omega
//    when the two tiles disagree on which side of the boundary one symbol instant falls →
//    drop the duplicate / insert the lost symbol (the warm-up's last symbol);
//  * output offsets: exclusive prefix sum of the adjusted counts.
struct rx_seam_result { unsigned long long total; unsigned rot_final, ndup, nmiss, nbad; float freq_tap; };   // freq_tap: rx_freq_tap(state) after the run

struct seam_step { unsigned insert, drop, k, bad; };

// gw(i): i-th symbol from the END of the current tile's (last) warm-up chunk, gp(i): i-th from the end of the previous
// tile's body; n_warm / n_prev: how many of each can be asked for.
template <typename INFO, typename GW, typename GP>
__device__ __forceinline__ seam_step seam_eval_core(const INFO &prev, const INFO &cur, float omega, int R, float quad,
                                                    int n_warm, int n_prev, GW gw, GP gp, const uint8_t *relabel) {
  seam_step r; r.insert = 0; r.drop = 0; r.bad = 0;
  const float d = cur.mu_begin - prev.mu_end;
  if (d > omega / 2 && cur.has_pre) r.insert = 1;
  else if (d < -omega / 2 && cur.count > 0) r.drop = 1;
  float dphi = fmodf(cur.phase_begin - prev.phase_end, 65536.0f);
  if (dphi < 0) dphi += 65536.0f;
  const int k = (int)floorf(dphi / quad + 0.5f);
  const float perr = fabsf(dphi - k * quad);
  float dm = fabsf(d);
  if (fabsf(dm - omega) < dm) dm = fabsf(dm - omega);
  if (perr > quad / 4 || dm > 0.5f) r.bad = 1;
  r.k = (unsigned)(k % R);
  // vote on the overlapping decisions: warm-up symbol w[i] (i = 0: last) ↔ body symbol p[i] of the previous tile,
  // shifted by one when the two tiles disagree on the boundary symbol (insert: w[0] is new; drop: body[0] repeats p[0])
  const int wo = r.insert ? 1 : 0, po = r.drop ? 1 : 0;
  const int have = min(n_warm - wo, n_prev - po);
  if (have >= kSeamVote) {
    int best = -1, best_k = 0;
    for (int kk = 0; kk < R; ++kk) {
      int hits = 0;
      for (int i = 0; i < kSeamVote; ++i) hits += relabel[kk * 256 + gw(wo + i)] == gp(po + i);
      if (hits > best) { best = hits; best_k = kk; }
    }
    if (best >= kSeamVote - 2) {                // clear majority: trust the decisions
      if ((unsigned)best_k != r.k) r.bad = 0;   // (the phase registers were ambiguous, the symbols are not)
      r.k = (unsigned)best_k;
      if (dm <= 0.5f) r.bad = 0;
    }
  }
  return r;
}

template <typename INFO, typename SYM>
__device__ __forceinline__ seam_step seam_eval(const INFO &prev, const INFO &cur, float omega, int R, float quad,
                                               const SYM *prev_body, const SYM *cur_warm, const uint8_t *relabel) {
  return seam_eval_core(prev, cur, omega, R, quad, (int)cur.n_warm, (int)prev.count,
                        [&](int i) { return rx_symbol_of(cur_warm[(int)cur.n_warm - 1 - i]); },
                        [&](int i) { return rx_symbol_of(prev_body[(int)prev.count - 1 - i]); }, relabel);
}

// Seam pass, two kernels, no single-workgroup scan:
//  k_rx_seam     one block per kSeamBlock consecutive tiles (256: four wavefronts, one per SIMD — a 1024-thread block needs
//                16 free wave slots on ONE CU at once and was seen waiting 0.3 ms for them next to the persistent fir kernel): evaluates the seams, block-local exclusive scan of
//                (symbol count, quadrant step) → fix[] holds block-local offsets, part[] the block totals;
//  k_rx_compact  one wavefront per kCompactTiles tiles: adds the (≤ a few dozen) preceding block totals, applies the seam
//                fix-ups and the quadrant relabelling while copying the tile's symbols to their final place.
//                Block 0 also leaves the run's totals in *res and rotates the carried carrier phase back into
//                the frame of tile 0, so the next queued run continues with the same symbol labelling.
constexpr unsigned kSeamBlock = 256, kSeamWaves = kSeamBlock / 64;
// k_rx_compact: consecutive tiles per (one-wavefront) workgroup.  One tile per workgroup was 34 944 workgroups to move 9 MB in the C2
// pipeline — 78 % of their wave cycles waiting, 59 µs next to fir_filter; a seam block (256 tiles) is a whole number of these groups.
constexpr unsigned kCompactTiles = 16;
static_assert(kSeamBlock % kCompactTiles == 0, "the tiles of a compaction workgroup share their seam block");
struct rx_seam_part { unsigned long long cnt; unsigned rot, ndup, nmiss, nbad; };

// block-local exclusive scan of (symbol count, quadrant step) over the kSeamBlock tiles of a workgroup → fix[], part[]
__device__ __forceinline__ void seam_block_scan(long long add, unsigned k, unsigned ins, unsigned drp, unsigned bad, unsigned j,
                                                unsigned n_tiles, unsigned rmask, rx_tile_fix *fix, rx_seam_part *part) {
  __shared__ unsigned long long s_cnt[kSeamWaves];
  __shared__ unsigned s_rot[kSeamWaves], s_d[kSeamWaves], s_m[kSeamWaves], s_b[kSeamWaves];
  const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  long long icnt = add;
  unsigned irot = k, nd = drp, nm = ins, nb = bad;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long long oc = __shfl_up(icnt, d, 64);
    const unsigned orot = __shfl_up(irot, d, 64);
    if (lane >= (unsigned)d) { icnt += oc; irot = (irot + orot) & rmask; }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { nd += __shfl_down(nd, d, 64); nm += __shfl_down(nm, d, 64); nb += __shfl_down(nb, d, 64); }
  if (lane == 63) { s_cnt[wv] = (unsigned long long)icnt; s_rot[wv] = irot; }
  if (lane == 0) { s_d[wv] = nd; s_m[wv] = nm; s_b[wv] = nb; }
  __syncthreads();
  unsigned long long woff = 0;
  unsigned wrot = 0;
  for (unsigned i = 0; i < wv; ++i) { woff += s_cnt[i]; wrot = (wrot + s_rot[i]) & rmask; }
  if (j < n_tiles) {
    rx_tile_fix f;
    f.out_offset = woff + (unsigned long long)(icnt - add);   // exclusive, block-local
    f.rot = (wrot + irot) & rmask;                            // inclusive, block-local
    f.drop_first = drp; f.insert_pre = ins;
    fix[j] = f;
  }
  if (tid == 0) {
    rx_seam_part p; p.cnt = 0; p.rot = 0; p.ndup = 0; p.nmiss = 0; p.nbad = 0;
    for (unsigned i = 0; i < kSeamWaves; ++i) { p.cnt += s_cnt[i]; p.rot = (p.rot + s_rot[i]) & rmask; p.ndup += s_d[i]; p.nmiss += s_m[i]; p.nbad += s_b[i]; }
    part[blockIdx.x] = p;
  }
}

template <typename INFO, typename SYM>
__device__ __forceinline__ void rx_seam_body(const INFO *info, rx_tile_fix *fix, unsigned n_tiles, float omega,
                                             int R, float quad, rx_seam_part *part, const SYM *stage, unsigned stage_stride,
                                             const SYM *wstage, unsigned wstride, const uint8_t *relabel) {
  const unsigned rmask = (unsigned)R - 1;   // nrotations is 2, 4 or 8 for every constellation (sdr.h:326-468)
  const unsigned j = blockIdx.x * kSeamBlock + threadIdx.x;
  long long add = 0;
  unsigned k = 0, ins = 0, drp = 0, bad = 0;
  if (j < n_tiles) {
    const INFO cur = info[j];
    add = (long long)cur.count;
    if (j > 0) {
      const seam_step st = seam_eval(info[j - 1], cur, omega, R, quad, stage + (unsigned long long)(j - 1) * stage_stride,
                                     wstage + (unsigned long long)j * wstride, relabel);
      add += (long long)st.insert - (long long)st.drop;
      k = st.k; ins = st.insert; drp = st.drop; bad = st.bad;
    }
  }
  seam_block_scan(add, k, ins, drp, bad, j, n_tiles, rmask, fix, part);
}
template <typename INFO, typename SYM>
__global__ __launch_bounds__(kSeamBlock) void k_rx_seam(const INFO *info, rx_tile_fix *fix, unsigned n_tiles, float omega,
                                                  int R, float quad, rx_seam_part *part, const SYM *stage, unsigned stage_stride,
                                                  const SYM *wstage, unsigned wstride, const uint8_t *relabel) {
  rx_seam_body<INFO, SYM>(info, fix, n_tiles, omega, R, quad, part, stage, stage_stride, wstage, wstride, relabel);
}

// relabel: [nrot][256] symbol relabelling per accumulated quadrant step; rx_relabel(sym, map) applies it to a record.
template <typename SYM, typename STATE>
__device__ __forceinline__ void rx_compact_body(const SYM *stage, unsigned stage_stride,
                                                const rx_tile_info_t<SYM> *info, const rx_tile_fix *fix, const rx_seam_part *part,
                                                const uint8_t *relabel /*[nrot][256]*/, unsigned n_tiles, int R, float quad,
                                                SYM *out, STATE *state, rx_seam_result *res) {
  const unsigned j0 = blockIdx.x * kCompactTiles;
  if (j0 >= n_tiles) return;
  const unsigned rmask = (unsigned)R - 1;
  const unsigned nparts = (n_tiles + kSeamBlock - 1) / kSeamBlock, mypart = j0 / kSeamBlock;
  // preceding block totals (lane-parallel, then wave-reduced; nparts is tiny) — once for the workgroup's tiles
  unsigned long long base = 0;
  unsigned brot = 0;
  for (unsigned i = threadIdx.x; i < mypart; i += 64) { base += part[i].cnt; brot += part[i].rot; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { base += __shfl_xor(base, d, 64); brot += __shfl_xor(brot, d, 64); }
  if (j0 == 0 && threadIdx.x == 0) {
    rx_seam_result sr; sr.total = 0; sr.rot_final = 0; sr.ndup = 0; sr.nmiss = 0; sr.nbad = 0; sr.freq_tap = rx_freq_tap(state);
    for (unsigned i = 0; i < nparts; ++i) {
      sr.total += part[i].cnt; sr.rot_final = (sr.rot_final + part[i].rot) & rmask;
      sr.ndup += part[i].ndup; sr.nmiss += part[i].nmiss; sr.nbad += part[i].nbad;
    }
    *res = sr;                 // host-pinned ring slot
    __threadfence_system();
    if (sr.rot_final) rx_rotate_back(state, sr.rot_final, quad);
  }
  // The workgroup's tiles side by side: lane t keeps tile j0 + t's records, every lane then has one symbol of EACH tile in flight (a
  // tile after the other — its records, its symbols, their relabelling, each a dependent load — took 75 µs next to fir_filter for the
  // C2 batch where one tile per workgroup had taken 59: the compaction is latency, not bytes).  A tile holds ≤ 64 body symbols at the
  // C2 geometry: one pass, a second one only where a tile has more.
  const unsigned nt = j0 + kCompactTiles < n_tiles ? kCompactTiles : n_tiles - j0;
  const unsigned lane = threadIdx.x;
  rx_tile_fix fl; fl.out_offset = 0; fl.rot = 0; fl.drop_first = 0; fl.insert_pre = 0;
  unsigned cnt_l = 0;
  if (lane < nt) { fl = fix[j0 + lane]; cnt_l = info[j0 + lane].count; }
  if (lane < nt && fl.insert_pre) {      // the warm-up's last symbol belongs to this tile
    const rx_tile_info_t<SYM> ti = info[j0 + lane];
    out[base + fl.out_offset] = rx_relabel(ti.pre, relabel + ((fl.rot + brot) & rmask) * 256);
  }
  unsigned maxc = cnt_l;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const unsigned o = __shfl_xor(maxc, d, 64); maxc = o > maxc ? o : maxc; }
  for (unsigned k0 = 0; k0 < maxc; k0 += 64) {
    SYM v[kCompactTiles];
    bool ok[kCompactTiles];
#pragma unroll
    for (unsigned t = 0; t < kCompactTiles; ++t) {
      const unsigned cnt = __shfl(cnt_l, t, 64), skip = __shfl(fl.drop_first, t, 64) ? 1u : 0u;
      const unsigned k = k0 + lane + skip;
      ok[t] = t < nt && k < cnt;
      if (ok[t]) v[t] = stage[(unsigned long long)(j0 + t) * stage_stride + k];
    }
#pragma unroll
    for (unsigned t = 0; t < kCompactTiles; ++t) {
      const unsigned long long off = __shfl(fl.out_offset, t, 64);
      const unsigned rot = __shfl(fl.rot, t, 64), ins = __shfl(fl.insert_pre, t, 64) ? 1u : 0u;
      if (ok[t]) out[base + off + ins + k0 + lane] = rx_relabel(v[t], relabel + ((rot + brot) & rmask) * 256);
    }
  }
}
template <typename SYM, typename STATE>
__global__ __launch_bounds__(64) void k_rx_compact(const SYM *stage, unsigned stage_stride,
                                                   const rx_tile_info_t<SYM> *info, const rx_tile_fix *fix, const rx_seam_part *part,
                                                   const uint8_t *relabel /*[nrot][256]*/, unsigned n_tiles, int R, float quad,
                                                   SYM *out, STATE *state, rx_seam_result *res) {
  rx_compact_body<SYM, STATE>(stage, stage_stride, info, fix, part, relabel, n_tiles, R, quad, out, state, res);
}


// ---- packed hard symbols ("hs2": QPSK decisions, 2 bits each) -------------------------------------------------------------
// The default leandvb chain feeds cstln_receiver's soft symbols to deconvol_sync, which reads nothing but `symbol & 3`
// (dvb.h:369-417).  In that chain the tiles keep only those two bits: 16 symbols per 32-bit word, MSB first (symbol k of a
// stream sits in word k/16 at bits 31−2(k%16) … 30−2(k%16)) — the order in which a tile shifts them in and in which the
// deconvolver's 64-bit window wants them (older symbols in higher bits).  Per tile: a row of packed body symbols (the last
// word left-aligned) and, in its info record, the last ≤ 16 warm-up and body symbols for the seam vote.
struct rx_tile_info_h {
  float mu_begin, phase_begin, mu_end, phase_end;
  unsigned count;                // body symbols
  unsigned has_pre;              // a warm-up symbol exists (pre = warm_tail & 3)
  unsigned n_warm;               // warm-up symbols held in warm_tail (≤ 16)
  unsigned warm_tail, body_tail; // newest symbol in bits 1:0
};

__device__ __forceinline__ unsigned hs2_get(const unsigned *words, long long k) {   // symbol k of a packed stream
  return (words[k >> 4] >> (30 - 2 * (int)(k & 15))) & 3u;
}
// apply a 4-entry symbol map (map4: new label of v in bits 2v+1:2v) to all 16 symbols of a word
__device__ __forceinline__ unsigned hs2_map_word(unsigned x, unsigned map4) {
  const unsigned M = 0x55555555u, b1 = (x >> 1) & M, b0 = x & M;
  const unsigned sel[4] = {~b1 & ~b0 & M, ~b1 & b0, b1 & ~b0, b1 & b0};
  unsigned out = 0;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const unsigned m = (map4 >> (2 * v)) & 3u;
    out |= (m & 1u ? sel[v] : 0u) | (m & 2u ? sel[v] << 1 : 0u);
  }
  return out;
}
__device__ __forceinline__ unsigned hs2_map4(const uint8_t *map) {   // [256]-entry relabel row → packed 4-entry map
  return (unsigned)(map[0] & 3) | ((unsigned)(map[1] & 3) << 2) | ((unsigned)(map[2] & 3) << 4) | ((unsigned)(map[3] & 3) << 6);
}
// The tiles' packed rows are stored TRANSPOSED: word w of tile j at hstage[w·pitch + j] (pitch = tiles, padded).  A wavefront
// of the tile kernel holds 64 consecutive tiles, its lanes reach "16 more symbols" within a few iterations of each other, so
// their 4-byte stores fall into the same 256 bytes and leave L2 as whole lines (row-major rows cost 5.7 bytes written per
// byte of payload: every lane dirtied its own line 4 bytes at a time).
// 16 symbols starting at symbol offset o (may be negative or reach past the row: zeros there) of tile column `col` (nw words)
__device__ __forceinline__ unsigned hs2_fetch16(const unsigned *col, unsigned long long pitch, long long nw, long long o) {
  const long long wi = o >> 4;                 // floor
  const int sh = 2 * (int)(o & 15);
  const unsigned a = (wi >= 0 && wi < nw) ? col[(unsigned long long)wi * pitch] : 0u;
  const unsigned b = (wi + 1 >= 0 && wi + 1 < nw) ? col[(unsigned long long)(wi + 1) * pitch] : 0u;
  return sh ? (a << sh) | (b >> (32 - sh)) : a;
}

__device__ __forceinline__ void rx_seam_h_body(const rx_tile_info_h *info, rx_tile_fix *fix, unsigned n_tiles, float omega,
                                               int R, float quad, rx_seam_part *part, const uint8_t *relabel) {
  const unsigned rmask = (unsigned)R - 1;
  const unsigned j = blockIdx.x * kSeamBlock + threadIdx.x;
  long long add = 0;
  unsigned k = 0, ins = 0, drp = 0, bad = 0;
  if (j < n_tiles) {
    const rx_tile_info_h cur = info[j];
    add = (long long)cur.count;
    if (j > 0) {
      const rx_tile_info_h prev = info[j - 1];
      const seam_step st = seam_eval_core(prev, cur, omega, R, quad, (int)cur.n_warm, (int)(prev.count < 16u ? prev.count : 16u),
                                          [&](int i) { return (cur.warm_tail >> (2 * i)) & 3u; },
                                          [&](int i) { return (prev.body_tail >> (2 * i)) & 3u; }, relabel);
      add += (long long)st.insert - (long long)st.drop;
      k = st.k; ins = st.insert; drp = st.drop; bad = st.bad;
    }
  }
  seam_block_scan(add, k, ins, drp, bad, j, n_tiles, rmask, fix, part);
}
__global__ __launch_bounds__(kSeamBlock) void k_rx_seam_h(const rx_tile_info_h *info, rx_tile_fix *fix, unsigned n_tiles, float omega,
                                                          int R, float quad, rx_seam_part *part, const uint8_t *relabel) {
  rx_seam_h_body(info, fix, n_tiles, omega, R, quad, part, relabel);
}

// Compaction of the packed tile columns: one LANE per tile (64 consecutive tiles per wavefront: the transposed staging is
// read coalesced) walks the output words that START inside its tile's symbol range [D, D + len) — word by word, so the lines
// of its stretch of the output fill up within a few loop trips and leave L2 whole.  A word that two tiles share is written by
// the later one, which takes the earlier tile's last symbols from that tile's column (no atomics, no pre-zeroing); the symbols
// before out_sym_offset in the first word (a caller's leftover symbols) are preserved.
// LPT lanes per tile (1, 2 or 4: a lane's trips are a chain of dependent loads, so a 4096-sample tile — 213 output words — is walked by
// four lanes, a quarter each; the wavefront then holds 64 / LPT consecutive tiles).
template <typename STATE, int LPT = 1>
__device__ __forceinline__ void rx_compact_h_body(const unsigned *hstage, unsigned long long pitch, const rx_tile_info_h *info,
                                                  const rx_tile_fix *fix, const rx_seam_part *part, const uint8_t *relabel,
                                                  unsigned n_tiles, int R, float quad, unsigned *out, unsigned long long out_sym_offset,
                                                  STATE *state, rx_seam_result *res) {
  const unsigned j = (blockIdx.x * 64u + threadIdx.x) / (unsigned)LPT, sub = threadIdx.x % (unsigned)LPT;
  const unsigned rmask = (unsigned)R - 1;
  const unsigned nparts = (n_tiles + kSeamBlock - 1) / kSeamBlock;
  if (j == 0 && sub == 0) {
    rx_seam_result sr; sr.total = 0; sr.rot_final = 0; sr.ndup = 0; sr.nmiss = 0; sr.nbad = 0; sr.freq_tap = rx_freq_tap(state);
    for (unsigned i = 0; i < nparts; ++i) {
      sr.total += part[i].cnt; sr.rot_final = (sr.rot_final + part[i].rot) & rmask;
      sr.ndup += part[i].ndup; sr.nmiss += part[i].nmiss; sr.nbad += part[i].nbad;
    }
    *res = sr;                 // host-pinned ring slot
    __threadfence_system();
    if (sr.rot_final) rx_rotate_back(state, sr.rot_final, quad);
  }
  // totals of the seam blocks before this wavefront's (kSeamBlock is a multiple of 64: one seam block per wavefront; only
  // the first tile's predecessor can sit in the block before)
  const unsigned mypart = (blockIdx.x * (64u / (unsigned)LPT)) / kSeamBlock;
  unsigned long long base = 0;
  unsigned brot = 0;
  for (unsigned i = threadIdx.x; i < mypart; i += 64) { base += part[i].cnt; brot += part[i].rot; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { base += __shfl_xor(base, d, 64); brot += __shfl_xor(brot, d, 64); }
  if (j >= n_tiles) return;
  unsigned long long pbase = base;
  unsigned pbrot = brot;
  if (j > 0 && (j % kSeamBlock) == 0) { pbase = base - part[mypart - 1].cnt; pbrot = brot - part[mypart - 1].rot; }   // (mod R below)
  // this tile's sequence T = [pre]? ++ body[skip …): destination symbols [D, D + len); body symbol b sits at b + Q
  const rx_tile_fix f = fix[j];
  const rx_tile_info_h ti = info[j];
  const long long skip = f.drop_first ? 1 : 0, ins = f.insert_pre ? 1 : 0;
  const long long D = (long long)(out_sym_offset + base + f.out_offset), len = ins + (long long)ti.count - skip;
  const long long Q = D + ins - skip;
  const unsigned map4 = hs2_map4(relabel + ((f.rot + brot) & rmask) * 256);
  const unsigned *col = hstage + j;
  const long long nw = ((long long)ti.count + 15) >> 4;
  // previous tile (for the shared first word)
  long long pQ = 0, pnw = 0, pD = 0, pins = 0;
  unsigned pmap4 = 0, ppre = 0;
  if (j > 0) {
    const rx_tile_fix pf = fix[j - 1];
    const rx_tile_info_h pti = info[j - 1];
    pins = pf.insert_pre ? 1 : 0;
    pD = (long long)(out_sym_offset + pbase + pf.out_offset);
    pQ = pD + pins - (pf.drop_first ? 1 : 0);
    pmap4 = hs2_map4(relabel + ((pf.rot + pbrot) & rmask) * 256);
    pnw = ((long long)pti.count + 15) >> 4;
    ppre = pti.warm_tail & 3u;
  }
  const long long end = D + len;
  const long long w_last = j == n_tiles - 1 ? (end + 15) >> 4 : end >> 4;  // exclusive; a partial last word belongs to the next tile
  auto one = [&](long long m) {
    const long long s0 = m << 4;                                            // first symbol of the word
    // part from this tile: symbols u with D ≤ s0+u < end
    unsigned cur = hs2_fetch16(col, pitch, nw, s0 - Q);
    if (ins && D >= s0 && D < s0 + 16) cur = (cur & ~(3u << (30 - 2 * (int)(D - s0)))) | ((ti.warm_tail & 3u) << (30 - 2 * (int)(D - s0)));
    const long long lo = D > s0 ? D - s0 : 0, hi = end < s0 + 16 ? end - s0 : 16;       // valid symbol slots [lo, hi)
    const unsigned mask = hi > lo ? (hi - lo >= 16 ? 0xffffffffu : ((1u << (2 * (int)(hi - lo))) - 1u) << (32 - 2 * (int)hi)) : 0u;
    unsigned word = hs2_map_word(cur, map4) & mask;
    if (lo > 0) {
      if (j > 0) {       // symbols [s0, D) are the previous tile's last ones
        unsigned pv = hs2_fetch16(col - 1, pitch, pnw, s0 - pQ);
        if (pins && pD >= s0 && pD < s0 + 16) pv = (pv & ~(3u << (30 - 2 * (int)(pD - s0)))) | (ppre << (30 - 2 * (int)(pD - s0)));
        word |= hs2_map_word(pv, pmap4) & (~0u << (32 - 2 * (int)lo));
      } else {           // tile 0: the caller's symbols before out_sym_offset stay
        word |= out[m] & (~0u << (32 - 2 * (int)lo));
      }
    }
    out[m] = word;
  };
  const long long m0 = D >> 4;
  if (sub == 0 && m0 < w_last) one(m0);   // the first word: may be shared with the previous tile, may hold the re-inserted symbol
  // Interior words — all 16 symbols from this tile's column, nothing to patch — four per trip with their five column words requested
  // together (a lane's trips are a chain of dependent loads otherwise: 213 round trips per 4096-sample tile, 1.3 ms for 8 captures),
  // the range cut into LPT stretches of whole trips.
  const long long m1 = m0 + 1, m_end = (end >> 4) > m1 ? (end >> 4) : m1;       // words [m1, m_end) lie wholly inside [D, end)
  constexpr int U = 8;                   // words per trip
  const long long span = ((m_end - m1 + LPT - 1) / LPT + (U - 1)) / U * U;
  long long m = m1 + (long long)sub * span;
  const long long stop = m + span < m_end ? m + span : m_end;
  {
    const int sh = 2 * (int)(((m1 << 4) - Q) & 15);
    for (; m + U <= stop; m += U) {
      const long long wi = ((m << 4) - Q) >> 4;
      unsigned w[U + 1];
#pragma unroll
      for (int i = 0; i <= U; ++i) w[i] = (wi + i < nw) ? col[(unsigned long long)(wi + i) * pitch] : 0u;
#pragma unroll
      for (int i = 0; i < U; ++i) out[m + i] = hs2_map_word(sh ? (w[i] << sh) | (w[i + 1] >> (32 - sh)) : w[i], map4);
    }
  }
  for (; m < stop; ++m) one(m);
  if (sub == LPT - 1) for (m = m_end; m < w_last; ++m) one(m);                  // (the last tile's partial word)
}
template <typename STATE>
__global__ __launch_bounds__(64) void k_rx_compact_h(const unsigned *hstage, unsigned long long pitch, const rx_tile_info_h *info,
                                                     const rx_tile_fix *fix, const rx_seam_part *part, const uint8_t *relabel,
                                                     unsigned n_tiles, int R, float quad, unsigned *out, unsigned long long out_sym_offset,
                                                     STATE *state, rx_seam_result *res) {
  rx_compact_h_body<STATE>(hstage, pitch, info, fix, part, relabel, n_tiles, R, quad, out, out_sym_offset, state, res);
}

#endif  // LSDR_RX_TILING_H
