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

template <typename INFO, typename SYM>
__device__ __forceinline__ seam_step seam_eval(const INFO &prev, const INFO &cur, float omega, int R, float quad,
                                               const SYM *prev_body, const SYM *cur_warm, const uint8_t *relabel) {
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
  const int have = min((int)cur.n_warm - wo, (int)prev.count - po);
  if (have >= kSeamVote) {
    int best = -1, best_k = 0;
    for (int kk = 0; kk < R; ++kk) {
      int hits = 0;
      for (int i = 0; i < kSeamVote; ++i) {
        const unsigned ws = rx_symbol_of(cur_warm[(int)cur.n_warm - 1 - wo - i]);
        const unsigned ps = rx_symbol_of(prev_body[(int)prev.count - 1 - po - i]);
        hits += relabel[kk * 256 + ws] == ps;
      }
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

// Seam pass, two kernels, no single-workgroup scan:
//  k_rx_seam     one block per kSeamBlock consecutive tiles (256: four wavefronts, one per SIMD — a 1024-thread block needs
//                16 free wave slots on ONE CU at once and was seen waiting 0.3 ms for them next to the persistent fir kernel): evaluates the seams, block-local exclusive scan of
//                (symbol count, quadrant step) → fix[] holds block-local offsets, part[] the block totals;
//  k_rx_compact  one wavefront per tile: adds the (≤ a few dozen) preceding block totals, applies the seam
//                fix-ups and the quadrant relabelling while copying the tile's symbols to their final place.
//                Block 0 also leaves the run's totals in *res and rotates the carried carrier phase back into
//                the frame of tile 0, so the next queued run continues with the same symbol labelling.
constexpr unsigned kSeamBlock = 256, kSeamWaves = kSeamBlock / 64;
struct rx_seam_part { unsigned long long cnt; unsigned rot, ndup, nmiss, nbad; };

template <typename INFO, typename SYM>
__global__ __launch_bounds__(kSeamBlock) void k_rx_seam(const INFO *info, rx_tile_fix *fix, unsigned n_tiles, float omega,
                                                  int R, float quad, rx_seam_part *part, const SYM *stage, unsigned stage_stride,
                                                  const SYM *wstage, unsigned wstride, const uint8_t *relabel) {
  const unsigned rmask = (unsigned)R - 1;   // nrotations is 2, 4 or 8 for every constellation (sdr.h:326-468)
  __shared__ unsigned long long s_cnt[kSeamWaves];
  __shared__ unsigned s_rot[kSeamWaves], s_d[kSeamWaves], s_m[kSeamWaves], s_b[kSeamWaves];
  const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned j = blockIdx.x * kSeamBlock + tid;
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

// relabel: [nrot][256] symbol relabelling per accumulated quadrant step; rx_relabel(sym, map) applies it to a record.
template <typename SYM, typename STATE>
__global__ __launch_bounds__(64) void k_rx_compact(const SYM *stage, unsigned stage_stride,
                                                   const rx_tile_info_t<SYM> *info, const rx_tile_fix *fix, const rx_seam_part *part,
                                                   const uint8_t *relabel /*[nrot][256]*/, unsigned n_tiles, int R, float quad,
                                                   SYM *out, STATE *state, rx_seam_result *res) {
  const unsigned j = blockIdx.x;
  if (j >= n_tiles) return;
  const unsigned rmask = (unsigned)R - 1;
  const unsigned nparts = (n_tiles + kSeamBlock - 1) / kSeamBlock, mypart = j / kSeamBlock;
  // preceding block totals (lane-parallel, then wave-reduced; nparts is tiny)
  unsigned long long base = 0;
  unsigned brot = 0;
  for (unsigned i = threadIdx.x; i < mypart; i += 64) { base += part[i].cnt; brot += part[i].rot; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { base += __shfl_xor(base, d, 64); brot += __shfl_xor(brot, d, 64); }
  if (j == 0 && threadIdx.x == 0) {
    rx_seam_result sr; sr.total = 0; sr.rot_final = 0; sr.ndup = 0; sr.nmiss = 0; sr.nbad = 0; sr.freq_tap = rx_freq_tap(state);
    for (unsigned i = 0; i < nparts; ++i) {
      sr.total += part[i].cnt; sr.rot_final = (sr.rot_final + part[i].rot) & rmask;
      sr.ndup += part[i].ndup; sr.nmiss += part[i].nmiss; sr.nbad += part[i].nbad;
    }
    *res = sr;                 // host-pinned ring slot
    __threadfence_system();
    if (sr.rot_final) rx_rotate_back(state, sr.rot_final, quad);
  }
  const rx_tile_fix f = fix[j];
  const rx_tile_info_t<SYM> ti = info[j];
  const uint8_t *map = relabel + ((f.rot + brot) & rmask) * 256;
  const SYM *src = stage + (unsigned long long)j * stage_stride;
  SYM *dst = out + base + f.out_offset;
  if (f.insert_pre) {
    if (threadIdx.x == 0) dst[0] = rx_relabel(ti.pre, map);
    dst += 1;
  }
  const unsigned skip = f.drop_first ? 1u : 0u;
  for (unsigned k = threadIdx.x + skip; k < ti.count; k += 64) {
    dst[k - skip] = rx_relabel(src[k], map);
  }
}


#endif  // LSDR_RX_TILING_H
