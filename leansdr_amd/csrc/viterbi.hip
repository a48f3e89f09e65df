// leansdr_amd/csrc/viterbi.hip — viterbi_sync (dvb.h:1173-1416) + viterbi_dec/trellis/bitpath (viterbi.h).
//
// 64-state trellis ↔ the 64 lanes of one wavefront: lane s owns path metric and path register of
// state s; predecessors' values arrive by wave shuffles (add-compare-select), the best state by a
// wave reduction.  The reference's arithmetic is kept exactly: int32 metrics, the labelled branch
// first with its (negative) cost then every branch with cost 0, `<=` so the last candidate wins,
// lowest-index best state, register-exchange paths of `depth` symbols.
//
// Parallelism comes from time tiling with verification.  A stream of chunks (128 FEC blocks each) is
// cut into tiles; tile 0 continues from the carried decoder state, every other tile starts
// `kWarm` chunks early from zero metrics.  Survivor paths merge within a few constraint lengths, after
// which metrics (relative to the best) and path registers no longer depend on the starting point:
// the tile then reproduces the sequential decoder bit for bit.  This is CHECKED, not assumed: the
// state of tile j after its warm-up is compared (64 normalised metrics + 64 path registers) with the
// end state of tile j−1 on the device; mismatching tiles are decoded again from their predecessor's end state (rounds,
// all in parallel), and a seam that still fails after six rounds makes the host re-decode from there sequentially.
// The decoders of the alignments not in force (they see the resync chunks only, dvb.h:1386-1410) are tiled and checked
// the same way and ride in the same launches as the main alignment's tiles.  Tiles get shorter when the input is short
// (more wavefronts on an otherwise idle chip).  Host↔device bookkeeping goes through the context's pinned arena.
//
// Work is skipped only where the reference's result cannot depend on it: the per-step "quality"
// (second-best − best) is needed on resync chunks only (dvb.h:1386-1394), metrics are renormalised
// once per chunk instead of every step (a common offset changes no comparison and cannot overflow in
// 128 steps), and the best-state reduction is skipped when all 64 path registers already agree on
// the output symbol.
#include "lsdr_internal.h"
#include <chrono>

namespace {

// LSDR_VIT_TIMING (debug hook): one stderr line per lsdr_viterbi_run call — wall time, number of launch → readback rounds and
// the time spent waiting in them.
struct vit_timing {
  bool on; double t0, wait_ms; int rounds; unsigned tiles, fixups;
  static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  vit_timing() : wait_ms(0), rounds(0), tiles(0), fixups(0) { static const bool e = getenv("LSDR_VIT_TIMING") != nullptr; on = e; t0 = on ? now() : 0; }
  ~vit_timing() { if (on && rounds) fprintf(stderr, "viterbi_run %.3f ms: %d rounds waited %.3f ms, %u tiles, %u fix-ups\n", now() - t0, rounds, wait_ms, tiles, fixups); }
};
#define VIT_SYNC(c, T) do { const double t__ = (T).on ? vit_timing::now() : 0; LSDR_TRY(lsdr_stage_sync(c)); if ((T).on) { (T).wait_ms += vit_timing::now() - t__; ++(T).rounds; } } while (0)

constexpr int kChunkBlocks = 128;   // viterbi_sync::chunk_size, dvb.h:1229
constexpr int kStates = 64;
static int vit_warm() {             // warm-up chunks of a tile (128 trellis steps each); LSDR_VIT_WARM is a tuning hook
  static int w = getenv("LSDR_VIT_WARM") ? atoi(getenv("LSDR_VIT_WARM")) : 4;
  return w < 1 ? 1 : w;
}
#define kWarm (vit_warm())
constexpr int kMaxSyncs = 64;

struct vit_code {           // fec_specs + typedefs, dvb.h:520-566,1179-1212
  int bits_in, bits_out, nus, ncs, nbits, depth, pathbits;
};

// trellis::init_convolutional (viterbi.h:59-92), branches sorted by coded symbol.  Branch-major / state-minor:
// the 64 lanes (= states) of one LDS read touch 64 consecutive bytes (state-major rows of 128 / 256 bytes put
// every lane on the same bank — a 32- to 64-way conflict on each of the ~6 table reads of a trellis step).
struct vit_tables {
  unsigned char pred[128][kStates];
  unsigned char us[128][kStates];
  unsigned char lab[128][kStates];       // coded symbol of branch k
  unsigned char by_label[256][kStates];  // branch index for a coded symbol, 255 = none
};

struct vit_state { int cost[kStates]; unsigned long long path[kStates]; };

struct vit_job {            // one wavefront's work
  unsigned long long first_chunk;    // first chunk to emit
  unsigned n_chunks;                 // chunks to emit
  unsigned warm;                     // warm-up chunks before first_chunk (0: start from init state)
  int sync;                          // alignment index
  int from_state;                    // index into states_in (warm == 0) or -1 (zero metrics)
  int emit;                          // write decoded bytes
  unsigned chunk_step;               // distance between emitted chunks (1: contiguous; P: the resync chunks only)
  unsigned slot;                     // where the job's begin/end/first states and totals go (normally its own index)
};

struct vit_args {
  const lsdr_softsymbol *in;
  unsigned char *out;
  const vit_tables *T;
  vit_code C;
  int bits_per_symbol, nshifts;
  int resync_phase0, resync_period;
  const unsigned char *maps;         // [nsyncs][256]
  const int *shifts;                 // [nsyncs]
  const vit_job *jobs;
  unsigned njobs;
  const vit_state *states_in;        // carried states, [nsyncs]
  vit_state *begin_states;           // [njobs] state at first_chunk (after warm-up)
  vit_state *end_states;             // [njobs] state after the job's last chunk
  vit_state *first_chunk_states;     // [njobs] state after the job's FIRST emitted chunk (for alignment switches)
  int *totals;                       // [njobs][n_chunks_max] totaldiscr per chunk (valid on resync chunks)
  unsigned totals_stride;
  vit_state *chunk_states;           // optional [njobs][totals_stride]: state after every chunk (sparse decoders)
  unsigned q4_n_main, q4_main_waves; // k_viterbi_q4: jobs [0, q4_n_main) fill the first q4_main_waves wavefronts, the rest the others
  // k_viterbi as the device-side repair round: only the jobs whose seam flag cond[slot] is set run, each from the end state of the slot
  // before its own (states_in = the slot array of end states), without warm-up; a job that starts a chain (from_state ≥ 0) has no seam
  const int *cond;
};

// Minimum over the 64 lanes, returned wave-uniform: DPP steps inside the rows of 16 (quad swaps, half mirror, mirror), two
// row broadcasts, one v_readlane — all on the vector ALU.  (The xor-butterfly of six dependent ds_bpermute it replaces cost
// ≈ 600 cycles; the best-state search runs whenever the survivors' oldest symbols disagree.)
__device__ __forceinline__ int wave_min(int v) {
#define LSDR_DPP_MIN(ctrl, row_mask) { const int o__ = __builtin_amdgcn_update_dpp(v, v, ctrl, row_mask, 0xf, false); v = o__ < v ? o__ : v; }
  LSDR_DPP_MIN(0xB1, 0xf)    // quad_perm [1,0,3,2]
  LSDR_DPP_MIN(0x4E, 0xf)    // quad_perm [2,3,0,1]
  LSDR_DPP_MIN(0x141, 0xf)   // row_half_mirror
  LSDR_DPP_MIN(0x140, 0xf)   // row_mirror: every lane of a row holds the row's minimum
  LSDR_DPP_MIN(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
  LSDR_DPP_MIN(0x143, 0xc)   // row_bcast:31 into rows 2 and 3: lane 63 holds the minimum of all
#undef LSDR_DPP_MIN
  return __builtin_amdgcn_readlane(v, 63);
}

// AND / OR over the 64 lanes (same DPP ladder as wave_min), returned wave-uniform
__device__ __forceinline__ void wave_and_or(unsigned v, unsigned &all_and, unsigned &all_or) {
  unsigned a = v, o = v;
#define LSDR_DPP_AO(ctrl, row_mask) { const unsigned xa__ = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)a, ctrl, row_mask, 0xf, false); \
                                       const unsigned xo__ = (unsigned)__builtin_amdgcn_update_dpp((int)o, (int)o, ctrl, row_mask, 0xf, false); a &= xa__; o |= xo__; }
  LSDR_DPP_AO(0xB1, 0xf)
  LSDR_DPP_AO(0x4E, 0xf)
  LSDR_DPP_AO(0x141, 0xf)
  LSDR_DPP_AO(0x140, 0xf)
  LSDR_DPP_AO(0x142, 0xa)
  LSDR_DPP_AO(0x143, 0xc)
#undef LSDR_DPP_AO
  all_and = (unsigned)__builtin_amdgcn_readlane((int)a, 63);
  all_or = (unsigned)__builtin_amdgcn_readlane((int)o, 63);
}

constexpr int kVitWaves = 4;
// One wavefront = one job.  Lane = trellis state.
// TWO = the code has two branches per state (rate 1/2: the headline mode): the lane's two predecessors, their input bits and
// the branch-by-label map live in registers, the two predecessors' metrics AND path registers are fetched together (six
// independent ds_bpermute) and the survivor is a select — one LDS round trip per trellis step instead of the generic
// path's chain of table read → shuffle → compare → table read → shuffle (≈ 900 cycles per step).  Same candidates, same
// order, same `<=` tie rule as the generic path: bit-identical.
// NUS = 2 / 4: that fast path (4: 8PSK 2/3 — four predecessors, eight labels, twelve independent ds_bpermute); 0 = generic.
template <int NUS>
__global__ __launch_bounds__(kVitWaves * 64) void k_viterbi(vit_args a) {
  constexpr bool TWO = NUS == 2;
  // The generic path walks the trellis tables in its inner loop: they live in LDS (≈ 41 KB per workgroup — three workgroups,
  // twelve wavefronts per CU).  The fast paths (NUS = 2, 4) only read their lane's constants once, straight from HBM: no LDS,
  // so the registers alone bound the occupancy (five wavefronts per SIMD) and tiles can be shorter for the same chip fill.
  __shared__ __attribute__((aligned(16))) char t_raw[NUS == 0 ? sizeof(vit_tables) : 16];
  const int lane = threadIdx.x & 63;
  const unsigned jid = blockIdx.x * kVitWaves + (threadIdx.x >> 6);   // one wavefront = one job; kVitWaves jobs share the LDS tables
  if (NUS == 0) {   // tables → LDS
    const unsigned *src = reinterpret_cast<const unsigned *>(a.T);
    unsigned *dst = reinterpret_cast<unsigned *>(t_raw);
    for (unsigned i = threadIdx.x; i < sizeof(vit_tables) / 4; i += kVitWaves * 64) dst[i] = src[i];
    __syncthreads();
  }
  const vit_tables &T = NUS == 0 ? *reinterpret_cast<const vit_tables *>(t_raw) : *a.T;
  if (jid >= a.njobs) return;
  vit_job job = a.jobs[jid];
  if (a.cond) {
    if (job.from_state >= 0 || !a.cond[job.slot]) return;
    job.warm = 0; job.from_state = (int)job.slot - 1;
  }
  const unsigned char *map = a.maps + job.sync * 256;
  const int shift = a.shifts[job.sync];
  const vit_code C = a.C;
  const unsigned long long pmask = C.pathbits == 64 ? ~0ull : ((1ull << C.pathbits) - 1);
  const int out_shift = (C.depth - 1) * C.nbits;
  const unsigned us_mask = (1u << C.nbits) - 1;
  const int discr_delay = 64 / C.bits_in;

  int cost;
  unsigned long long path;
  if (job.from_state >= 0) { cost = a.states_in[job.from_state].cost[lane]; path = a.states_in[job.from_state].path[lane]; }
  else { cost = 0; path = 0; }
  // TWO: per-lane constants of the trellis
  const int pred0 = T.pred[0][lane], pred1 = T.pred[1][lane];
  const unsigned us0 = T.us[0][lane], us1 = T.us[1][lane];
  const unsigned bl4 = (unsigned)T.by_label[0][lane] | ((unsigned)T.by_label[1][lane] << 8) | ((unsigned)T.by_label[2][lane] << 16) |
                       ((unsigned)T.by_label[3][lane] << 24);
  // NUS == 4: predecessors 2, 3 and the labels 4..7
  const int pred2 = T.pred[2][lane], pred3 = T.pred[3][lane];
  const unsigned us2 = T.us[2][lane], us3 = T.us[3][lane];
  const unsigned bl4h = (unsigned)T.by_label[4][lane] | ((unsigned)T.by_label[5][lane] << 8) | ((unsigned)T.by_label[6][lane] << 16) |
                        ((unsigned)T.by_label[7][lane] << 24);

  for (long long q = -(long long)job.warm; q < (long long)job.n_chunks; ++q) {
    const unsigned long long c = (unsigned long long)((long long)job.first_chunk + q * (long long)job.chunk_step);
    const bool emitting = q >= 0;
    if (q == 0) {   // (metrics are normalised at every chunk boundary)
      a.begin_states[job.slot].cost[lane] = cost;
      a.begin_states[job.slot].path[lane] = path;
    }
    const bool resync = ((c + (unsigned long long)a.resync_phase0) % (unsigned)a.resync_period) == 0;
    const bool want_q = resync && emitting;
    int total = 0;
    unsigned long long outstream = 0;
    int nout = 0;
    unsigned char *pout = a.out + c * (unsigned)(kChunkBlocks * C.bits_in / 8);
    const lsdr_softsymbol *pin = a.in + c * (unsigned)(kChunkBlocks * a.nshifts) + shift;
    // update_sync (dvb.h:1353-1364) for the whole chunk at once: lane l prepares FEC blocks l and l+64 (coded
    // symbol = mapped bits of the block's `nshifts` symbols, cost = their summed costs); the trellis loop below
    // then takes block b's pair with two v_readlane instead of dependent global loads on its critical path.
    unsigned my_cs[2];
    int my_cost[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const lsdr_softsymbol *pb = pin + (unsigned)(lane + 64 * h) * (unsigned)a.nshifts;
      unsigned cs = 0;
      int cst = 0;
      for (int i = 0; i < a.nshifts; ++i) {
        const lsdr_softsymbol ss = pb[i];
        cs = ((cs << a.bits_per_symbol) | map[ss.symbol]) & 0xffu;
        cst += ss.cost;
      }
      my_cs[h] = cs; my_cost[h] = cst;
    }
    bool bulk = false;     // TWO: the outputs of the current group of 16 steps have been emitted at its first step
    for (int b = 0; b < kChunkBlocks; ++b) {
      const unsigned cs1 = (unsigned)__builtin_amdgcn_readlane((int)(b < 64 ? my_cs[0] : my_cs[1]), b & 63);
      const int cost1 = __builtin_amdgcn_readlane(b < 64 ? my_cost[0] : my_cost[1], b & 63);
      // viterbi_dec::update(nm = 1), viterbi.h:202-260
      int best_m = 0x7fffffff, bk = 0;
      if (TWO) {
        // rate 1/2: 32-symbol paths of one bit each (the 64-bit register's upper half stays 0).  The two predecessors' metrics
        // come first; the survivor's path is then fetched from the ONE lane that won (three ds_bpermute per step instead of
        // six; the path exchange is off the metric recurrence's critical chain).
        const int c0 = __shfl(cost, pred0, 64), c1 = __shfl(cost, pred1, 64);
        const unsigned k1 = (bl4 >> (8u * (cs1 & 3u))) & 255u;          // branch carrying the received label, 255 = none
        if (k1 != 255u) { best_m = (k1 ? c1 : c0) + cost1; bk = (int)k1; }   // first candidate: always ≤ the initial maximum
        if (c0 <= best_m) { best_m = c0; bk = 0; }
        if (c1 <= best_m) { best_m = c1; bk = 1; }
        const unsigned ps = __shfl((unsigned)path, bk ? pred1 : pred0, 64);
        path = (unsigned long long)((ps << 1) | (bk ? us1 : us0));       // nbits = 1, pathbits = 32
        cost = best_m;
      } else if (NUS == 4) {
        // four predecessors: their metrics first, then the path register of the ONE that won (six ds_bpermute, not twelve)
        const int c0 = __shfl(cost, pred0, 64), c1 = __shfl(cost, pred1, 64), c2 = __shfl(cost, pred2, 64), c3 = __shfl(cost, pred3, 64);
        const unsigned k1 = ((cs1 & 4u ? bl4h : bl4) >> (8u * (cs1 & 3u))) & 255u;   // branch carrying the received label, 255 = none
        if (k1 != 255u) { best_m = (k1 == 0 ? c0 : k1 == 1 ? c1 : k1 == 2 ? c2 : c3) + cost1; bk = (int)k1; }
        if (c0 <= best_m) { best_m = c0; bk = 0; }
        if (c1 <= best_m) { best_m = c1; bk = 1; }
        if (c2 <= best_m) { best_m = c2; bk = 2; }
        if (c3 <= best_m) { best_m = c3; bk = 3; }
        const int bp = bk == 0 ? pred0 : bk == 1 ? pred1 : bk == 2 ? pred2 : pred3;
        const unsigned slo = __shfl((unsigned)path, bp, 64), shi = __shfl((unsigned)(path >> 32), bp, 64);
        const unsigned usel = bk == 0 ? us0 : bk == 1 ? us1 : bk == 2 ? us2 : us3;
        path = ((((unsigned long long)shi << 32 | slo) << C.nbits) | usel) & pmask;
        cost = best_m;
      } else {
        {
          const unsigned k1 = T.by_label[cs1][lane];
          const unsigned kk = k1 == 255 ? 0u : k1;
          const int pc = __shfl(cost, (int)T.pred[kk][lane], 64);
          if (k1 != 255) { const int m = pc + cost1; if (m <= best_m) { best_m = m; bk = (int)k1; } }
        }
        for (int k = 0; k < C.nus; ++k) {
          const int m = __shfl(cost, (int)T.pred[k][lane], 64);
          if (m <= best_m) { best_m = m; bk = k; }
        }
        const int bp = T.pred[bk][lane];
        const unsigned lo = __shfl((unsigned)path, bp, 64), hi = __shfl((unsigned)(path >> 32), bp, 64);
        path = ((((unsigned long long)hi << 32 | lo) << C.nbits) | T.us[bk][lane]) & pmask;
        cost = best_m;
      }
      // output symbol of the best state (lowest index among the minima); skip the search when all agree.  Everything from
      // here to the store is wave-uniform (scalar registers): sym_u, the bit accumulator, the counters.
      if (TWO && !want_q) {
        // Rate 1/2, no quality wanted on this chunk: the symbol the reference returns at step t+j is bit 31 of the best path
        // at t+j, i.e. the decision of step t+j−31 — bit 31−j of the time-t path of that survivor's ancestor.  Where ALL 64
        // survivors of time t agree on bits 31…16, the outputs of steps t … t+15 are those bits whichever state is best then:
        // one AND/OR reduction per 16 steps instead of an agreement test (and a possible best-state search) per step.
        if (!(emitting && job.emit)) continue;            // nothing leaves this job on this chunk
        if ((b & 15) == 0) {
          unsigned all_and, all_or;
          wave_and_or((unsigned)path, all_and, all_or);
          bulk = ((all_and ^ all_or) >> 16) == 0u;
          if (bulk) {
            outstream = (outstream << 16) | (all_and >> 16);
            nout += 16;
            if (nout >= 32) {
              if (lane == 0) *reinterpret_cast<unsigned *>(pout) = __builtin_bswap32((unsigned)(outstream >> (nout - 32)));
              pout += 4;
              nout -= 32;
            }
          }
        }
        if (bulk) continue;
      }
      const unsigned sym_out = (unsigned)(path >> out_shift) & us_mask;
      unsigned sym_u = 0;
      if (emitting || want_q) {
        const unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)sym_out);   // lane 0 is active: every lane of a job is
        const bool all_same = __all(sym_out == s0);
        sym_u = s0;
        if (!all_same || (want_q && b >= discr_delay)) {
          const int best_tpm = wave_min(cost);
          const unsigned long long mask = __ballot(cost == best_tpm);
          const int best_state = __ffsll((long long)mask) - 1;
          sym_u = (unsigned)__builtin_amdgcn_readlane((int)sym_out, best_state);
          if (want_q && b >= discr_delay) {
            // second-best in the reference's scan = 2nd smallest with multiplicity (viterbi.h:246-251)
            const int second = __popcll(mask) > 1 ? best_tpm : wave_min(cost == best_tpm ? 0x7fffffff : cost);
            total += second - best_tpm;
          }
        }
      }
      if (emitting && job.emit) {
        // a chunk emits 128·bits_in bits = a whole number of 32-bit words, and its output starts 16·bits_in bytes into the
        // stream: whole aligned words, first decoded bit in the first byte's MSB
        outstream = (outstream << C.bits_in) | sym_u;
        nout += C.bits_in;
        if (nout >= 32) {
          if (lane == 0) *reinterpret_cast<unsigned *>(pout) = __builtin_bswap32((unsigned)(outstream >> (nout - 32)));
          pout += 4;
          nout -= 32;
        }
      }
    }
    // renormalise once per chunk (the reference subtracts the best metric after every step)
    cost -= wave_min(cost);
    if (emitting) {
      const unsigned ci = (unsigned)q;
      if (lane == 0) a.totals[(size_t)job.slot * a.totals_stride + ci] = total;
      if (a.chunk_states) {
        vit_state *st = a.chunk_states + (size_t)job.slot * a.totals_stride + ci;
        st->cost[lane] = cost; st->path[lane] = path;
      }
      if (ci == 0 && a.first_chunk_states) { a.first_chunk_states[job.slot].cost[lane] = cost; a.first_chunk_states[job.slot].path[lane] = path; }
    }
  }
  a.end_states[job.slot].cost[lane] = cost;
  a.end_states[job.slot].path[lane] = path;
}

// ---------------------------------------------------------------------------------------------------------------------
// Rates 1/2 (QPSK) and 2/3 (8PSK) of the K = 7, 171/133 code: FOUR lanes per tile, sixteen tiles per wavefront.
//
// Lane q of a quad holds the sixteen states 16q … 16q+15 of its tile in registers (metric + path each).  The predecessors of
// state s' = 16q + r (viterbi.h:59-92) sit in fixed registers of a fixed lane of the same quad:
//   rate 1/2: 2(s' & 31) + e, e = 0, 1, input bit s' >> 5: registers 2(r & 7) + e of lane 2(q & 1) + (r >> 3);
//   rate 2/3: 4(s' & 15) + e, e = 0 … 3, input bits s' >> 4 (reversed): registers 4(r & 3) + e of lane r >> 2;
// i.e. quad permutes with compile-time register indices (DPP operands of the adds and moves: no LDS, no shuffle latency).
// One wavefront instruction advances sixteen tiles, where the lane = state kernel above advances one and waits for two LDS
// round trips per step.
//
// Same arithmetic as viterbi_dec::update (viterbi.h:202-260), restated so that it needs no branch per state:
//   * candidates in the reference's order are [labelled branch + cost, then every branch by ascending coded symbol] with
//     `<=`, so the last of equal candidates wins.  The labelled branch reappears with cost 0, so a POSITIVE cost can never
//     win: cost' = min(cost, 0) gives the same survivor and metric.  With cost' < 0 the plain copy of the labelled branch
//     never wins either, and (labelled + cost') loses a tie against any other branch (they all come later); among plain
//     branches the one with the highest coded symbol wins a tie.
//   * metrics are kept × 16.  The four low bits carry the tie rule: the number of branches into the state with a higher
//     coded symbol (0 … 3), or 4 on the labelled branch when cost' < 0; after the minimum they are cleared.  No two
//     candidates of a state are equal then, and a strict compare of the sums IS the reference's choice.  Range: |cost| ≤ 32768
//     per step, renormalised every 128 steps, spread ≤ 6·32768: × 16 stays inside ± 2^27.
//   * the code is linear: the coded symbol of branch e into state 16q + r is Lr(r) ^ Le(e) ^ Lq(q), and the coded symbols of
//     the branches into one state form a coset of {Le(e)}, so the tie rank of a branch depends on its coded symbol alone.
//     The received symbol is XORed with Lq once per step; Lr(r) ^ Le(e) then picks one of NCS per-step addends by a
//     compile-time register index.
// lsdr_viterbi_create checks these properties on the trellis tables it built before this kernel is ever chosen.
namespace q4 {
constexpr unsigned kG1 = 0171, kG2 = 0133;
constexpr unsigned par8(unsigned x) { return (x ^ (x >> 1) ^ (x >> 2) ^ (x >> 3) ^ (x >> 4) ^ (x >> 5) ^ (x >> 6) ^ (x >> 7)) & 1u; }
// coded symbol of the branch into state s from its predecessor number e (shift register = predecessor | input bits << 6)
template <int NUS> constexpr unsigned branch_label(unsigned s, unsigned e) {
  if (NUS == 2) { const unsigned reg = ((s & 31u) << 1) | e | ((s >> 5) << 6); return (par8(reg & kG1) << 1) | par8(reg & kG2); }
  const unsigned reg = ((s & 15u) << 2) | e | ((s >> 4) << 6);
  return (par8(reg & kG1) << 2) | (par8(reg & kG2) << 1) | par8(reg & (kG2 << 1));
}
// the input symbol stored in the path register of state s (trellis::us: the reversed bits are the state's top bits)
template <int NUS> constexpr unsigned state_us(unsigned s) { return NUS == 2 ? s >> 5 : (((s >> 4) & 1u) << 1) | (s >> 5); }

template <int NUS> struct regs { int c[16]; unsigned p[16]; unsigned ph[NUS == 4 ? 16 : 1]; };

template <int CTRL> __device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ unsigned dppu(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int quad_min(int v) { int o = dpp<0xB1>(v); v = o < v ? o : v; o = dpp<0x4E>(v); return o < v ? o : v; }

// rate 1/2: add-compare-select for the eight states r = 8·HI + j
template <int HI>
__device__ __forceinline__ void acs2_half(const regs<2> &R, regs<2> &N, const int (&A)[4], unsigned us31) {
  constexpr int CTRL = HI ? 0xDD : 0x88;   // quad_perm [1,3,1,3] / [0,2,0,2]
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = 8 * HI + j;
    const int me = dpp<CTRL>(R.c[2 * j]) + A[branch_label<2>((unsigned)r, 0)];
    const int mo = dpp<CTRL>(R.c[2 * j + 1]) + A[branch_label<2>((unsigned)r, 1)];
    const bool odd = mo < me;
    N.c[r] = (odd ? mo : me) & ~15;
    const unsigned pe = dppu<CTRL>(R.p[2 * j]), po = dppu<CTRL>(R.p[2 * j + 1]);
    N.p[r] = __builtin_amdgcn_alignbit(odd ? po : pe, us31, 31);   // (survivor's path << 1) | input bit of the state
  }
}
// rate 2/3: the four states r = 4·K + j, predecessors in lane K
template <int K>
__device__ __forceinline__ void acs4_quarter(const regs<4> &R, regs<4> &N, const int (&A)[8], unsigned us) {
  constexpr int CTRL = K * 0x55;           // quad_perm [K,K,K,K]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 4 * K + j;
    const int m0 = dpp<CTRL>(R.c[4 * j]) + A[branch_label<4>((unsigned)r, 0)], m1 = dpp<CTRL>(R.c[4 * j + 1]) + A[branch_label<4>((unsigned)r, 1)];
    const int m2 = dpp<CTRL>(R.c[4 * j + 2]) + A[branch_label<4>((unsigned)r, 2)], m3 = dpp<CTRL>(R.c[4 * j + 3]) + A[branch_label<4>((unsigned)r, 3)];
    const bool o01 = m1 < m0, o23 = m3 < m2;
    const int l01 = o01 ? m1 : m0, l23 = o23 ? m3 : m2;
    const bool hi = l23 < l01;
    N.c[r] = (hi ? l23 : l01) & ~15;
    const unsigned a0 = dppu<CTRL>(R.p[4 * j]), a1 = dppu<CTRL>(R.p[4 * j + 1]), a2 = dppu<CTRL>(R.p[4 * j + 2]), a3 = dppu<CTRL>(R.p[4 * j + 3]);
    const unsigned b0 = dppu<CTRL>(R.ph[4 * j]), b1 = dppu<CTRL>(R.ph[4 * j + 1]), b2 = dppu<CTRL>(R.ph[4 * j + 2]), b3 = dppu<CTRL>(R.ph[4 * j + 3]);
    const unsigned a01 = o01 ? a1 : a0, a23 = o23 ? a3 : a2, b01 = o01 ? b1 : b0, b23 = o23 ? b3 : b2;
    const unsigned lo = hi ? a23 : a01, hw = hi ? b23 : b01;
    N.ph[r] = __builtin_amdgcn_alignbit(hw, lo, 29);               // 64-bit (path << 3) | input symbol of the state
    N.p[r] = (lo << 3) | us;
  }
}

// one trellis step of every tile of the wavefront.  d: this lane's decoded symbol register holding the step's symbol in
// lane SRC of the quad (16·cost' | coded symbol, or 8: nothing to add); lq = Lq(q); tl[l]: tie rank of coded symbol l ^ Lq(q).
template <int NUS, int SRC>
__device__ __forceinline__ void step(regs<NUS> &R, int d, unsigned lq, const int (&tl)[2 * NUS], unsigned us) {
  const int S = dpp<SRC * 0x55>(d);                 // quad_perm [SRC,SRC,SRC,SRC]
  const unsigned csx = ((unsigned)S ^ lq) & 15u;
  const int hotv = (S & ~15) | NUS;
  int A[2 * NUS];
#pragma unroll
  for (int l = 0; l < 2 * NUS; ++l) A[l] = csx == (unsigned)l ? hotv : tl[l];
  regs<NUS> N;
  if constexpr (NUS == 2) { acs2_half<0>(R, N, A, us); acs2_half<1>(R, N, A, us); }
  else { acs4_quarter<0>(R, N, A, us); acs4_quarter<1>(R, N, A, us); acs4_quarter<2>(R, N, A, us); acs4_quarter<3>(R, N, A, us); }
  R = N;
}

// best metric of the tile (× 16), in every lane of the quad
template <int NUS> __device__ __forceinline__ int tile_min(const regs<NUS> &R) {
  int m = R.c[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) m = R.c[r] < m ? R.c[r] : m;
  return quad_min(m);
}

// oldest path symbol of the best state, lowest state index among equal metrics (viterbi.h:232-260).  The key is
// metric·2^(5+B) | state·2^B | symbol (B = 1, 3): 16·metric ≥ −2^26 (128 steps of −32768 after a renormalisation) keeps
// metric·512 ≥ −2^31.
template <int NUS> __device__ __forceinline__ unsigned best_symbol(const regs<NUS> &R, int q) {
  constexpr int B = NUS == 2 ? 1 : 3;
  int k = 0x7fffffff;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const unsigned old = NUS == 2 ? R.p[r] : R.ph[r] << 1;     // oldest symbol in the top B bits
    const int key = (int)__builtin_amdgcn_alignbit((unsigned)(R.c[r] | r), old, 32 - B);
    k = key < k ? key : k;
  }
  constexpr unsigned low = (16u << B) - 1u;
  k = (int)((((unsigned)k & ~low) << 2) | ((unsigned)q << (4 + B)) | ((unsigned)k & low));
  return (unsigned)quad_min(k) & (NUS == 2 ? 1u : 3u);
}

// second smallest − smallest metric of the tile, equal minima counted separately (viterbi.h:246-251)
template <int NUS> __device__ __forceinline__ int quality(const regs<NUS> &R) {
  int m1 = R.c[0] < R.c[1] ? R.c[0] : R.c[1], m2 = R.c[0] < R.c[1] ? R.c[1] : R.c[0];
#pragma unroll
  for (int r = 2; r < 16; ++r) {
    const int x = R.c[r];
    const int lo = m1 < x ? m1 : x, hi = m1 < x ? x : m1;
    m2 = hi < m2 ? hi : m2;
    m1 = lo;
  }
  {
    const int o1 = dpp<0xB1>(m1), o2 = dpp<0xB1>(m2);
    const int hi = m1 < o1 ? o1 : m1, lo2 = m2 < o2 ? m2 : o2;
    m1 = m1 < o1 ? m1 : o1; m2 = hi < lo2 ? hi : lo2;
  }
  {
    const int o1 = dpp<0x4E>(m1), o2 = dpp<0x4E>(m2);
    const int hi = m1 < o1 ? o1 : m1, lo2 = m2 < o2 ? m2 : o2;
    m1 = m1 < o1 ? m1 : o1; m2 = hi < lo2 ? hi : lo2;
  }
  return (m2 - m1) >> 4;
}

// Where ALL 64 survivors of a tile agree on their oldest symbols, those are the decoder's next outputs whichever state is
// best then: the output of step t + j is the oldest symbol of the best path at t + j, i.e. symbol number depth − 1 − j of
// its ancestor's path at t.  Rate 1/2: 16 one-bit symbols (path bits 31…16); rate 2/3: 8 symbols of the 21 (path bits
// 62…39), two bits each.  Returns agreement; `bits` = the 16 output bits, first one in bit 15.
template <int NUS> __device__ __forceinline__ bool agreed_outputs(const regs<NUS> &R, unsigned &bits) {
  unsigned aa, oo;
  if constexpr (NUS == 2) { aa = R.p[0]; oo = R.p[0]; } else { aa = R.ph[0]; oo = R.ph[0]; }
#pragma unroll
  for (int r = 1; r < 16; ++r) { const unsigned x = NUS == 2 ? R.p[r] : R.ph[r]; aa &= x; oo |= x; }
  aa &= dppu<0xB1>(aa); oo |= dppu<0xB1>(oo);
  aa &= dppu<0x4E>(aa); oo |= dppu<0x4E>(oo);
  if constexpr (NUS == 2) { bits = aa >> 16; return ((aa ^ oo) >> 16) == 0u; }
  else {
    unsigned b = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) b = (b << 2) | ((aa >> (28 - 3 * j)) & 3u);   // path bits 62−3j … 60−3j live in ph bits 30−3j … 28−3j
    bits = b;
    return ((aa ^ oo) & 0x7fffff80u) == 0u;
  }
}

template <int NUS> __device__ __forceinline__ void load_state(regs<NUS> &R, const vit_state &st, int q) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    R.c[r] = st.cost[16 * q + r] << 4; R.p[r] = (unsigned)st.path[16 * q + r];
    if constexpr (NUS == 4) R.ph[r] = (unsigned)(st.path[16 * q + r] >> 32);
  }
}
template <int NUS> __device__ __forceinline__ void store_state(const regs<NUS> &R, vit_state &st, int q) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    st.cost[16 * q + r] = R.c[r] >> 4;
    st.path[16 * q + r] = NUS == 4 ? ((unsigned long long)R.ph[r] << 32) | R.p[r] : (unsigned long long)R.p[r];
  }
}

struct sym4 { unsigned v[4]; };
}  // namespace q4

template <int NUS>
__global__ __launch_bounds__(64) void k_viterbi_q4(vit_args a) {
  using namespace q4;
  constexpr int BITS_IN = NUS == 2 ? 1 : 2;          // decoded bits per step
  constexpr int NCS = 2 * NUS;
  const int lane = threadIdx.x, q = lane & 3;
  const unsigned quad = (unsigned)lane >> 2;
  // jobs [0, q4_n_main) are the current alignment's tiles, the rest the other alignments' (they want the quality of every
  // step and emit nothing): the two kinds never share a wavefront
  unsigned jid;
  bool valid;
  if (blockIdx.x < a.q4_main_waves) { jid = blockIdx.x * 16u + quad; valid = jid < a.q4_n_main; }
  else { jid = a.q4_n_main + (blockIdx.x - a.q4_main_waves) * 16u + quad; valid = jid < a.njobs; }
  vit_job job;
  if (valid) job = a.jobs[jid];
  else { job.first_chunk = 0; job.n_chunks = 0; job.warm = 0; job.sync = 0; job.from_state = -1; job.emit = 0; job.chunk_step = 1; job.slot = 0; }
  const int wmax = -wave_min(-(int)job.warm), nmax = -wave_min(-(int)job.n_chunks);
  const unsigned char *map = a.maps + job.sync * 256;
  const unsigned map_lo = (unsigned)map[0] | ((unsigned)map[1] << 8) | ((unsigned)map[2] << 16) | ((unsigned)map[3] << 24);
  const unsigned map_hi = (unsigned)map[4] | ((unsigned)map[5] << 8) | ((unsigned)map[6] << 16) | ((unsigned)map[7] << 24);
  const unsigned lq = branch_label<NUS>(16u * (unsigned)q, 0);
  // the states' input symbol (trellis::us) as the step functions want it: rate 1/2 in bit 31, rate 2/3 as the low bits
  const unsigned us = NUS == 2 ? state_us<2>(16u * (unsigned)q) << 31 : state_us<4>(16u * (unsigned)q);
  int tl[NCS];   // tie rank of the branch whose coded symbol is l ^ Lq(q): the number of higher coded symbols in its coset
#pragma unroll
  for (int l = 0; l < NCS; ++l) {
    const unsigned y = (unsigned)l ^ lq;
    int n = 0;
#pragma unroll
    for (int e = 1; e < NUS; ++e) n += (y ^ branch_label<NUS>(0, (unsigned)e)) > y ? 1 : 0;
    tl[l] = n;
  }

  regs<NUS> R;
  if (valid && job.from_state >= 0) load_state<NUS>(R, a.states_in[job.from_state], q);
  else {
#pragma unroll
    for (int r = 0; r < 16; ++r) { R.c[r] = 0; R.p[r] = 0; if constexpr (NUS == 4) R.ph[r] = 0; }
  }
  // the lane's four symbols of a group of sixteen steps: symbols 16g + 4i + q of the chunk (step 16g + 4i + k reads register
  // i of lane k of the quad)
  auto load_group = [&](long long qq, int g, sym4 &o) {
    const unsigned long long c = (unsigned long long)((long long)job.first_chunk + qq * (long long)job.chunk_step);
    const unsigned *p = reinterpret_cast<const unsigned *>(a.in + c * (unsigned)kChunkBlocks + (unsigned)(16 * g + q));
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = p[4 * i];
  };
  sym4 nxt = {{0, 0, 0, 0}};
  if (valid && job.n_chunks) load_group(-(long long)job.warm, 0, nxt);

  for (int qq = -wmax; qq < nmax; ++qq) {
    const bool act = valid && qq >= -(int)job.warm && qq < (int)job.n_chunks;
    if (!act) continue;
    const unsigned long long c = (unsigned long long)((long long)job.first_chunk + (long long)qq * (long long)job.chunk_step);
    const bool emitting = qq >= 0;
    if (qq == 0) store_state<NUS>(R, a.begin_states[job.slot], q);
    const bool resync = ((c + (unsigned long long)a.resync_phase0) % (unsigned)a.resync_period) == 0;
    const bool want_q = resync && emitting, do_emit = emitting && job.emit;
    int total = 0;
    unsigned outw = 0;
    unsigned *pout = reinterpret_cast<unsigned *>(a.out + c * (unsigned)(kChunkBlocks * BITS_IN / 8));
#pragma unroll 1
    for (int g = 0; g < 8; ++g) {
      // decode this group's symbols (update_sync, dvb.h:1353-1364, nshifts = 1), fetch the next group's
      int d[4];
      const sym4 cur = nxt;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned raw = cur.v[i];
        const unsigned sym = (raw >> 16) & 255u;
        const unsigned cs = sym < 4u ? (map_lo >> (8u * sym)) & 255u : sym < 8u ? (map_hi >> (8u * (sym - 4u))) & 255u : (unsigned)map[sym];
        const int cost = (int)(short)(raw & 0xffffu);
        // 16·cost' with the coded symbol in the low bits; 8 where nothing is added to any branch (cost ≥ 0, or a coded
        // symbol no branch carries)
        d[i] = (cs < (unsigned)NCS && cost < 0) ? (cost * 16) | (int)cs : 8;
      }
      {
        const bool last = g == 7;
        long long nq = last ? (long long)qq + 1 : (long long)qq;
        if (nq >= (long long)job.n_chunks) nq = (long long)job.n_chunks - 1;
        load_group(nq, last ? 0 : g + 1, nxt);
      }
      bool bulk = false;
#pragma unroll 1
      for (int jj = 0; jj < 4; ++jj) {
        const int dj = d[0];
        d[0] = d[1]; d[1] = d[2]; d[2] = d[3];
        // a check covers the outputs of the next 16 (rate 1/2) / 8 (rate 2/3) steps
        const bool check = NUS == 2 ? jj == 0 : (jj & 1) == 0;
#define LSDR_Q4_AFTER(first)                                                                               \
        if (do_emit) {                                                                                     \
          if (first) {                                                                                     \
            unsigned bits;                                                                                 \
            bulk = agreed_outputs<NUS>(R, bits);                                                           \
            if (bulk) outw = (outw << 16) | bits;                                                          \
          }                                                                                                \
          if (!bulk) outw = (outw << BITS_IN) | best_symbol<NUS>(R, q);                                    \
        }                                                                                                  \
        if (want_q && g >= 4 / BITS_IN) total += quality<NUS>(R);
        step<NUS, 0>(R, dj, lq, tl, us); LSDR_Q4_AFTER(check)
        step<NUS, 1>(R, dj, lq, tl, us); LSDR_Q4_AFTER(false)
        step<NUS, 2>(R, dj, lq, tl, us); LSDR_Q4_AFTER(false)
        step<NUS, 3>(R, dj, lq, tl, us); LSDR_Q4_AFTER(false)
#undef LSDR_Q4_AFTER
      }
      // sixteen steps make 16·BITS_IN bits: a word every other group (rate 1/2) / every group (rate 2/3)
      if (do_emit && (NUS == 4 || (g & 1))) { if (q == 0) pout[NUS == 4 ? g : g >> 1] = __builtin_bswap32(outw); }
    }
    // renormalise once per chunk (the reference subtracts the best metric after every step)
    {
      const int m = tile_min<NUS>(R);
#pragma unroll
      for (int r = 0; r < 16; ++r) R.c[r] -= m;
    }
    if (emitting) {
      const unsigned ci = (unsigned)qq;
      if (q == 0) a.totals[(size_t)job.slot * a.totals_stride + ci] = total;
      if (a.chunk_states) store_state<NUS>(R, a.chunk_states[(size_t)job.slot * a.totals_stride + ci], q);
      if (ci == 0 && a.first_chunk_states) store_state<NUS>(R, a.first_chunk_states[job.slot], q);
    }
  }
  if (valid) store_state<NUS>(R, a.end_states[job.slot], q);
}

// seam check: begin state of job j (j ≥ 1) == end state of job j−1
__global__ __launch_bounds__(64) void k_vit_verify(const vit_state *begin_states, const vit_state *end_states, unsigned njobs,
                                                   int *bad) {
  const unsigned j = blockIdx.x + 1;
  if (j >= njobs) return;
  const int lane = threadIdx.x;
  const bool same = begin_states[j].cost[lane] == end_states[j - 1].cost[lane] &&
                    begin_states[j].path[lane] == end_states[j - 1].path[lane];
  if (!__all(same) && lane == 0) bad[j] = 1;
}

}  // namespace

struct lsdr_viterbi {
  lsdr_ctx *ctx;
  vit_code C;
  int cstln, rate, bits_per_symbol, nshifts, nsyncs;
  size_t dbg_chunks = 0;   // chunks consumed so far (LSDR_VIT_DEBUG only)
  int current_sync, resync_phase, resync_period;
  std::vector<unsigned char> maps;   // [nsyncs][256]
  std::vector<int> shifts;
  vit_tables *d_T;
  unsigned char *d_maps;
  int *d_shifts;
  std::vector<vit_state> states;     // host mirror of the carried decoder states [nsyncs]
  vit_state *d_states;
  // scratch
  vit_job *d_jobs; vit_state *d_begin, *d_end, *d_first, *d_chunk; int *d_totals, *d_bad;
  vit_state *d_fix;                   // explicit start states of fix-up jobs
  size_t jobs_cap, totals_cap, chunk_cap, fix_cap, first_cap;
  unsigned last_tiles, last_bad;
  unsigned long long dev_repaired = 0, host_rounds = 0;   // seams re-decoded by the device-side round / launch → readback rounds the host had to add (lsdr_viterbi_repair_stats)
  size_t budget_chunks;               // chunks attempted per call: shrinks after an alignment switch, regrows
  bool q4;                            // rate 1/2 QPSK / 2/3 8PSK may use the four-lanes-per-tile kernel (trellis structure checked at create)
  bool q4_call;                       // ... and the current lsdr_viterbi_run call does
  unsigned warm_others;               // warm-up chunks of the other alignments' tiles (grows when their seams fail)
  unsigned char *d_bounce = nullptr;  // 4-byte aligned stand-in for a misaligned `out` (lsdr_viterbi_run)
  size_t bounce_cap = 0;
};

static int vit_code_for(int rate, vit_code *c, const unsigned short **polys) {
  static const unsigned short G1 = 0171, G2 = 0133;
  static const unsigned short p12[] = {G1, G2};
  static const unsigned short p23[] = {G1, G2, (unsigned short)(G2 << 1)};
  static const unsigned short p46[] = {G1, G2, (unsigned short)(G2 << 1), (unsigned short)(G1 << 2), (unsigned short)(G2 << 2), (unsigned short)(G2 << 3)};
  static const unsigned short p34[] = {G1, G2, (unsigned short)(G2 << 1), (unsigned short)(G1 << 2)};
  static const unsigned short p45[] = {G1, G2, (unsigned short)(G2 << 1), (unsigned short)(G1 << 2), (unsigned short)(G1 << 3)};
  static const unsigned short p56[] = {G1, G2, (unsigned short)(G2 << 1), (unsigned short)(G1 << 2), (unsigned short)(G2 << 3), (unsigned short)(G1 << 4)};
  static const unsigned short p78[] = {G1, G2, (unsigned short)(G2 << 1), (unsigned short)(G2 << 2), (unsigned short)(G2 << 3), (unsigned short)(G1 << 4), (unsigned short)(G2 << 5), (unsigned short)(G1 << 6)};
  switch (rate) {   // {bits_in, bits_out, NUS, NCS, NBITS, DEPTH, path width}
    case LSDR_FEC12: *c = {1, 2, 2, 4, 1, 32, 32}; *polys = p12; return 0;
    case LSDR_FEC23: *c = {2, 3, 4, 8, 3, 21, 64}; *polys = p23; return 0;
    case LSDR_FEC46: *c = {4, 6, 16, 64, 4, 16, 64}; *polys = p46; return 0;
    case LSDR_FEC34: *c = {3, 4, 8, 16, 3, 21, 64}; *polys = p34; return 0;
    case LSDR_FEC45: *c = {4, 5, 16, 32, 4, 16, 64}; *polys = p45; return 0;
    case LSDR_FEC56: *c = {5, 6, 32, 64, 5, 12, 64}; *polys = p56; return 0;
    case LSDR_FEC78: *c = {7, 8, 128, 256, 7, 9, 64}; *polys = p78; return 0;
  }
  return -1;
}

// trellis::init_convolutional (viterbi.h:59-92), branches of each state ordered by coded symbol.  Host only.
static int vit_build_trellis(const vit_code &C, const unsigned short *polys, vit_tables *T) {
  memset(T, 0, sizeof(*T));
  memset(T->by_label, 255, sizeof(T->by_label));
  int nG = 0;
  while ((1 << nG) < C.ncs) ++nG;
  struct br { int cs, pred, us; };
  std::vector<br> all[kStates];
  for (int s = 0; s < kStates; ++s)
    for (int us = 0; us < C.nus; ++us) {
      unsigned long long reg = (unsigned long long)s;
      int us_rev = 0;
      for (int b = 1; b < C.nus; b *= 2) if (us & b) us_rev |= (C.nus / 2 / b);
      reg |= (unsigned long long)us_rev * kStates;
      unsigned cs = 0;
      for (int g = 0; g < nG; ++g) cs = (cs << 1) | (unsigned)__builtin_parityll(reg & polys[g]);
      reg /= (unsigned)C.nus;
      all[reg].push_back({(int)cs, s, us});
    }
  for (int s = 0; s < kStates; ++s) {
    std::vector<br> &L = all[s];
    for (size_t i = 0; i < L.size(); ++i) for (size_t j = i + 1; j < L.size(); ++j) if (L[j].cs < L[i].cs) { br t = L[i]; L[i] = L[j]; L[j] = t; }
    if ((int)L.size() != C.nus) return -1;
    for (int k = 0; k < C.nus; ++k) {
      T->pred[k][s] = (unsigned char)L[k].pred; T->us[k][s] = (unsigned char)L[k].us; T->lab[k][s] = (unsigned char)L[k].cs;
      T->by_label[L[k].cs][s] = (unsigned char)k;
    }
  }
  return 0;
}

// k_viterbi_q4's assumptions, checked on the tables of the code at hand: the predecessors of state s are NUS·(s mod 64/NUS) + e,
// all its branches store the same input symbol, branch e carries q4::branch_label(s, e), coded symbols are linear in (s & 15),
// (s & 48) and e, and the branches are ordered by coded symbol.  Host only.
static bool vit_q4_fits(const vit_code &C, int nshifts, const vit_tables *T) {
  bool ok = ((C.nus == 2 && C.bits_out == 2 && C.nbits == 1 && C.depth == 32) || (C.nus == 4 && C.bits_out == 3 && C.nbits == 3 && C.depth == 21)) &&
            nshifts == 1;
  for (int s = 0; ok && s < kStates; ++s) {
    for (int k = 0; ok && k < C.nus; ++k) {
      const unsigned e = (unsigned)T->pred[k][s] % (unsigned)C.nus;
      const unsigned lab = C.nus == 2 ? q4::branch_label<2>((unsigned)s, e) : q4::branch_label<4>((unsigned)s, e);
      const unsigned lin = C.nus == 2 ? q4::branch_label<2>((unsigned)s & 15u, e) ^ q4::branch_label<2>((unsigned)s & 48u, 0)
                                      : q4::branch_label<4>((unsigned)s & 15u, e) ^ q4::branch_label<4>((unsigned)s & 48u, 0);
      const unsigned us = C.nus == 2 ? q4::state_us<2>((unsigned)s) : q4::state_us<4>((unsigned)s);
      ok = T->pred[k][s] == (unsigned)C.nus * ((unsigned)s % (unsigned)(kStates / C.nus)) + e && T->us[k][s] == us && T->lab[k][s] == lab &&
           lab == lin && (k == 0 || T->lab[k][s] > T->lab[k - 1][s]);
    }
    for (int cs = 0; ok && cs < 256; ++cs) {
      int want = 255;
      for (int k = 0; k < C.nus; ++k) if (T->lab[k][s] == cs) want = k;
      ok = T->by_label[cs][s] == want;
    }
  }
  return ok;
}

static int vit_launch(lsdr_viterbi *v, const lsdr_softsymbol *in, uint8_t *out, const std::vector<vit_job> &jobs,
                      unsigned stride, bool chunk_states, int phase0, const std::vector<vit_state> *start_states = nullptr,
                      const vit_state *dev_start_states = nullptr, bool keep_slots = false, size_t n_slots = 0,
                      bool write_first = true, size_t n_main = (size_t)-1, vit_args *args_out = nullptr) {
  lsdr_ctx *c = v->ctx;
  const size_t nj = keep_slots ? (n_slots > jobs.size() ? n_slots : jobs.size()) : jobs.size();   // capacity of the slot arrays
  if (start_states && v->fix_cap < start_states->size()) {
    (void)hipFree(v->d_fix);
    LSDR_HIP(hipMalloc((void **)&v->d_fix, start_states->size() * sizeof(vit_state)));
    v->fix_cap = start_states->size();
  }
  if (v->jobs_cap < nj) {
    (void)hipFree(v->d_jobs); (void)hipFree(v->d_begin); (void)hipFree(v->d_end); (void)hipFree(v->d_bad);
    LSDR_HIP(hipMalloc((void **)&v->d_jobs, nj * sizeof(vit_job)));
    LSDR_HIP(hipMalloc((void **)&v->d_begin, nj * sizeof(vit_state)));
    LSDR_HIP(hipMalloc((void **)&v->d_end, nj * sizeof(vit_state)));
    LSDR_HIP(hipMalloc((void **)&v->d_bad, 2 * nj * sizeof(int)));   // [0, nj): the current seam flags; [nj, 2·nj): those the device-side repair round acted on
    v->jobs_cap = nj;
  }
  // d_first belongs to the current alignment's tiles: it is read lazily after the other alignments have been launched,
  // so their launches neither write nor reallocate it.
  if (write_first && v->first_cap < nj) {
    (void)hipFree(v->d_first);
    LSDR_HIP(hipMalloc((void **)&v->d_first, nj * sizeof(vit_state)));
    v->first_cap = nj;
  }
  if (v->totals_cap < nj * stride) {
    (void)hipFree(v->d_totals);
    LSDR_HIP(hipMalloc((void **)&v->d_totals, nj * stride * sizeof(int)));
    v->totals_cap = nj * stride;
  }
  if (chunk_states && v->chunk_cap < nj * stride) {
    (void)hipFree(v->d_chunk);
    LSDR_HIP(hipMalloc((void **)&v->d_chunk, nj * stride * sizeof(vit_state)));
    v->chunk_cap = nj * stride;
  }
  std::vector<vit_job> up(jobs);
  if (!keep_slots) for (size_t i = 0; i < up.size(); ++i) up[i].slot = (unsigned)i;
  LSDR_TRY(lsdr_stage_h2d(c, v->d_jobs, up.data(), up.size() * sizeof(vit_job)));
  LSDR_TRY(lsdr_stage_h2d(c, v->d_states, v->states.data(), v->nsyncs * sizeof(vit_state)));
  if (start_states)
    LSDR_TRY(lsdr_stage_h2d(c, v->d_fix, start_states->data(), start_states->size() * sizeof(vit_state)));
  vit_args a;
  a.in = in; a.out = out; a.T = v->d_T; a.C = v->C;
  a.bits_per_symbol = v->bits_per_symbol; a.nshifts = v->nshifts;
  a.resync_phase0 = phase0; a.resync_period = v->resync_period;
  a.maps = v->d_maps; a.shifts = v->d_shifts;
  a.jobs = v->d_jobs; a.states_in = dev_start_states ? dev_start_states : (start_states ? v->d_fix : v->d_states);
  a.begin_states = v->d_begin; a.end_states = v->d_end; a.first_chunk_states = write_first ? v->d_first : nullptr;
  a.totals = v->d_totals; a.totals_stride = stride;
  a.chunk_states = chunk_states ? v->d_chunk : nullptr;
  a.njobs = (unsigned)up.size();
  a.q4_n_main = a.njobs; a.q4_main_waves = 0;
  a.cond = nullptr;
  if (args_out) *args_out = a;
  const dim3 grid((unsigned)((up.size() + kVitWaves - 1) / kVitWaves)), block(kVitWaves * 64);
  // (test hooks, read at every call so that one process can exercise every kernel)
  const bool generic_only = getenv("LSDR_VIT_GENERIC") != nullptr;   // test hook: every mode through the generic path
  const bool lane_only = getenv("LSDR_VIT_LANE") != nullptr;         // test hook: rates 1/2 and 2/3 on the lane = state kernel
  // k_viterbi_q4 takes as long as its longest wavefront however few tiles there are; a launch of a few tiles (a fix-up round,
  // a sequential re-run) is a latency problem and goes to the lane = state kernel, which walks one tile five times faster
  if (v->q4_call && !generic_only && !lane_only && (up.size() >= 1024 || getenv("LSDR_VIT_Q4") != nullptr)) {
    a.q4_n_main = (unsigned)(n_main < up.size() ? n_main : up.size());
    a.q4_main_waves = (a.q4_n_main + 15u) / 16u;
    const unsigned other_waves = (a.njobs - a.q4_n_main + 15u) / 16u;
    if (a.C.nus == 2) hipLaunchKernelGGL(k_viterbi_q4<2>, dim3(a.q4_main_waves + other_waves), dim3(64), 0, c->stream, a);
    else hipLaunchKernelGGL(k_viterbi_q4<4>, dim3(a.q4_main_waves + other_waves), dim3(64), 0, c->stream, a);
  } else if (a.C.nus == 2 && a.C.bits_out == 2 && !generic_only) hipLaunchKernelGGL(k_viterbi<2>, grid, block, 0, c->stream, a);
  else if (a.C.nus == 4 && a.C.bits_out == 3 && !generic_only) hipLaunchKernelGGL(k_viterbi<4>, grid, block, 0, c->stream, a);
  else hipLaunchKernelGGL(k_viterbi<0>, grid, block, 0, c->stream, a);
  LSDR_HIP(hipGetLastError());
  return LSDR_OK;
}

extern "C" {

int lsdr_viterbi_create(lsdr_ctx *c, int cstln, int rate, lsdr_viterbi **out) {
  LSDR_ARG(c && out);
  vit_code C;
  const unsigned short *polys = nullptr;
  if (vit_code_for(rate, &C, &polys)) { lsdr_set_error("viterbi_sync: CR not supported"); return LSDR_E_UNSUPPORTED; }
  lsdr::cstln_tables tab;
  if (lsdr::build_cstln(cstln, rate, tab) < 0) { lsdr_set_error("viterbi_sync: constellation/code rate not supported"); return LSDR_E_ARG; }
  int bps = 0;
  while ((1 << (bps + 1)) <= tab.nsymbols) ++bps;   // log2i(nsymbols)
  if (bps * (C.bits_out / bps) != C.bits_out) { lsdr_set_error("viterbi_sync: code rate not suitable for this constellation"); return LSDR_E_ARG; }
  LSDR_HIP(hipSetDevice(c->device));
  lsdr_viterbi *v = new lsdr_viterbi();
  v->ctx = c; v->C = C; v->cstln = cstln; v->rate = rate;
  v->bits_per_symbol = bps;
  const int nconj = tab.nsymbols == 2 ? 1 : 2;                                   // dvb.h:1248-1252
  const int nrot = (tab.nsymbols == 2 || tab.nsymbols == 4) ? tab.nrotations / 2 : tab.nrotations;   // dvb.h:1254-1266
  v->nshifts = C.bits_out / bps;
  v->nsyncs = nconj * nrot * v->nshifts;
  if (v->nsyncs > kMaxSyncs) { delete v; lsdr_set_error("viterbi_sync: too many alignments"); return LSDR_E_UNSUPPORTED; }
  v->current_sync = 0; v->resync_phase = 0; v->resync_period = 32;
  v->maps.assign((size_t)v->nsyncs * 256, 0);
  v->shifts.assign(v->nsyncs, 0);
  for (int s = 0; s < v->nsyncs; ++s) {   // [shift|conj|rot], init_map dvb.h:1336-1351
    const int rot = s % nrot, conj = (s / nrot) % nconj, shift = s / nrot / nconj;
    v->shifts[s] = shift;
    const float angle = (float)(2 * M_PI * rot / tab.nrotations);
    const float ca = cosf(angle), sa = sinf(angle);
    for (int i = 0; i < tab.nsymbols; ++i) {
      int8_t I = tab.symbols[i][0], Q = tab.symbols[i][1];
      if (conj) Q = (int8_t)-Q;
      const int8_t RI = (int8_t)(I * ca - Q * sa);
      const int8_t RQ = (int8_t)(I * sa + Q * ca);
      v->maps[(size_t)s * 256 + i] = tab.symbol[(size_t)(uint8_t)RI * 256 + (uint8_t)RQ];
    }
  }
  vit_tables *T = new vit_tables();
  if (vit_build_trellis(C, polys, T)) { delete T; delete v; lsdr_set_error("viterbi_sync: invalid convolutional code"); return LSDR_E_ARG; }
  v->q4 = vit_q4_fits(C, v->nshifts, T);
  LSDR_HIP(hipMalloc((void **)&v->d_T, sizeof(vit_tables)));
  LSDR_HIP(hipMemcpy(v->d_T, T, sizeof(vit_tables), hipMemcpyHostToDevice));
  delete T;
  LSDR_HIP(hipMalloc((void **)&v->d_maps, v->maps.size()));
  LSDR_HIP(hipMemcpy(v->d_maps, v->maps.data(), v->maps.size(), hipMemcpyHostToDevice));
  LSDR_HIP(hipMalloc((void **)&v->d_shifts, v->nsyncs * sizeof(int)));
  LSDR_HIP(hipMemcpy(v->d_shifts, v->shifts.data(), v->nsyncs * sizeof(int), hipMemcpyHostToDevice));
  v->states.assign(v->nsyncs, vit_state());
  for (auto &s : v->states) memset(&s, 0, sizeof(s));
  LSDR_HIP(hipMalloc((void **)&v->d_states, v->nsyncs * sizeof(vit_state)));
  v->d_jobs = nullptr; v->d_begin = v->d_end = v->d_first = v->d_chunk = nullptr; v->d_totals = nullptr; v->d_bad = nullptr;
  v->d_fix = nullptr; v->fix_cap = 0;
  v->jobs_cap = v->totals_cap = v->chunk_cap = v->first_cap = 0;
  v->last_tiles = v->last_bad = 0;
  v->budget_chunks = (size_t)1 << 40;
  v->q4_call = false;
  v->warm_others = 8;
  *out = v;
  return LSDR_OK;
}

int lsdr_viterbi_q4_supported(int cstln, int rate) {
  vit_code C;
  const unsigned short *polys = nullptr;
  if (vit_code_for(rate, &C, &polys)) return -1;
  lsdr::cstln_tables tab;
  if (lsdr::build_cstln(cstln, rate, tab) < 0) return -1;
  int bps = 0;
  while ((1 << (bps + 1)) <= tab.nsymbols) ++bps;
  if (bps * (C.bits_out / bps) != C.bits_out) return -1;
  vit_tables *T = new vit_tables();
  const int bad = vit_build_trellis(C, polys, T);
  const bool ok = !bad && vit_q4_fits(C, C.bits_out / bps, T);
  delete T;
  return bad ? -1 : ok ? 1 : 0;
}

void lsdr_viterbi_destroy(lsdr_viterbi *v) {
  if (!v) return;
  (void)hipStreamSynchronize(v->ctx->stream);
  (void)hipFree(v->d_T); (void)hipFree(v->d_maps); (void)hipFree(v->d_shifts); (void)hipFree(v->d_states);
  (void)hipFree(v->d_jobs); (void)hipFree(v->d_begin); (void)hipFree(v->d_end); (void)hipFree(v->d_first);
  (void)hipFree(v->d_chunk); (void)hipFree(v->d_totals); (void)hipFree(v->d_bad); (void)hipFree(v->d_fix);
  (void)hipFree(v->d_bounce);
  delete v;
}

int lsdr_viterbi_set_resync_period(lsdr_viterbi *v, int p) { LSDR_ARG(v && p >= 1); v->resync_period = p; return LSDR_OK; }
int lsdr_viterbi_current_sync(const lsdr_viterbi *v) { return v ? v->current_sync : -1; }
int lsdr_viterbi_repair_stats(const lsdr_viterbi *v, unsigned long long *device_repaired, unsigned long long *host_rounds) {
  LSDR_ARG(v);
  if (device_repaired) *device_repaired = v->dev_repaired;
  if (host_rounds) *host_rounds = v->host_rounds;
  return LSDR_OK;
}
int lsdr_viterbi_stats(const lsdr_viterbi *v, unsigned *tiles, unsigned *bad) {
  LSDR_ARG(v);
  if (tiles) *tiles = v->last_tiles;
  if (bad) *bad = v->last_bad;
  return LSDR_OK;
}

static int viterbi_run_aligned(lsdr_viterbi *v, const lsdr_softsymbol *in, size_t n_in, uint8_t *out, size_t cap_out,
                               size_t *consumed, size_t *produced);

// The kernels store decoded bytes four at a time (32-bit words).  A caller's `out` may sit at any byte offset — a pipebuf<u8>
// write pointer after compaction and mpeg_sync's byte-granular reads — so a pointer that is not 4-byte aligned is served through
// an aligned bounce buffer and one device copy (same bytes; rare).
int lsdr_viterbi_run(lsdr_viterbi *v, const lsdr_softsymbol *in, size_t n_in, uint8_t *out, size_t cap_out,
                     size_t *consumed, size_t *produced) {
  LSDR_ARG(v && consumed && produced);
  if (((uintptr_t)out & 3u) == 0 || !out || !cap_out) return viterbi_run_aligned(v, in, n_in, out, cap_out, consumed, produced);
  lsdr_ctx *c = v->ctx;
  LSDR_HIP(hipSetDevice(c->device));
  // (sized for what this run can produce — at most one byte per input symbol — not for the caller's whole capacity)
  const size_t need = cap_out < n_in + 64 ? cap_out : n_in + 64;
  if (v->bounce_cap < need) {
    LSDR_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(v->d_bounce);
    v->d_bounce = nullptr; v->bounce_cap = 0;
    LSDR_HIP(hipMalloc((void **)&v->d_bounce, need + 64));
    v->bounce_cap = need;
  }
  const int rc = viterbi_run_aligned(v, in, n_in, v->d_bounce, need, consumed, produced);
  if (rc) return rc;
  if (*produced) {
    LSDR_HIP(hipMemcpyAsync(out, v->d_bounce, *produced, hipMemcpyDeviceToDevice, c->stream));
    LSDR_HIP(hipStreamSynchronize(c->stream));      // the aligned path has waited for its results too: `out` is complete on return
  }
  return LSDR_OK;
}

static int viterbi_run_aligned(lsdr_viterbi *v, const lsdr_softsymbol *in, size_t n_in, uint8_t *out, size_t cap_out,
                               size_t *consumed, size_t *produced) {
  LSDR_ARG(v && consumed && produced);
  *consumed = 0; *produced = 0;
  vit_timing vt;
  const vit_code &C = v->C;
  const size_t sym_per_chunk = (size_t)v->nshifts * kChunkBlocks;
  const size_t bytes_per_chunk = (size_t)C.bits_in * kChunkBlocks / 8;
  if (n_in < sym_per_chunk + (size_t)(v->nshifts - 1)) return LSDR_OK;   // dvb.h:1372-1373
  size_t chunks = (n_in - (size_t)(v->nshifts - 1)) / sym_per_chunk;
  if (chunks > cap_out / bytes_per_chunk) chunks = cap_out / bytes_per_chunk;
  if (!chunks) return LSDR_OK;
  LSDR_ARG(in && out);
  // Everything decoded after an alignment switch is thrown away (the new alignment has to redo it), so while the
  // decoder keeps switching (no lock) a call only looks one resync period ahead; the look-ahead doubles again
  // with every switch-free call.  The output does not depend on how the stream is cut into calls.
  if (chunks > v->budget_chunks) chunks = v->budget_chunks;
  lsdr_ctx *c = v->ctx;
  LSDR_HIP(hipSetDevice(c->device));

  const int P = v->resync_period, phase0 = v->resync_phase;
  // resync chunks of this call: c with (c + phase0) % P == 0
  std::vector<unsigned long long> rs;
  for (unsigned long long cc = (unsigned long long)((P - phase0) % P); cc < chunks; cc += (unsigned)P) rs.push_back(cc);

  // ---- jobs.  Current alignment: tiles of TL chunks (each starting kWarm chunks early) so that every
  // resync chunk is the FIRST chunk of a tile; other alignments: one sequential job each over the resync chunks.
  // Tile length (chunks): a tile is one wavefront walking (kWarm + TL)·128 trellis steps, kWarm of them thrown away, and the
  // output does not depend on the tiling.  Long inputs take the longest tile that divides the resync period (≤ 32: a fifth
  // of the work is warm-up at 16, a third at 8); short inputs take shorter tiles, which put more wavefronts on an otherwise
  // idle chip.
  unsigned TL = (unsigned)P;
  if (P >= 8) { TL = 8; while (TL < 32 && P % (int)(TL * 2) == 0) TL *= 2; }
  bool q4_kernel = false;
  {
    static const int forced = getenv("LSDR_VIT_TL") ? atoi(getenv("LSDR_VIT_TL")) : 0;   // tuning hook
    static const int want_x2 = getenv("LSDR_VIT_WANT") ? atoi(getenv("LSDR_VIT_WANT")) : 0;   // tuning hook: wavefronts per SIMD, in halves
    const bool lane_kernel = getenv("LSDR_VIT_LANE") != nullptr || getenv("LSDR_VIT_GENERIC") != nullptr;
    const bool q4_always = getenv("LSDR_VIT_Q4") != nullptr;            // test hook: k_viterbi_q4 whatever the input length
    // Which kernel.  k_viterbi_q4 carries sixteen tiles per wavefront and is bound by instruction issue (a wavefront alone on
    // its SIMD already issues back to back; more wavefronts add nothing): what counts is the number of trellis steps and an
    // even spread over the SIMDs, so it wants long tiles (an eighth of the work is warm-up at 32 chunks) and about one
    // wavefront on every other SIMD (6144 tiles on 256 CUs) before it shortens them.  A call too short for that — fewer
    // than 8 chunks per such tile — is a latency problem instead: the lane = state kernel, one tile per wavefront and 1.5
    // wavefronts per SIMD, finishes it sooner (4 Mi symbols: 0.77 ms against 1.2; 32 Mi: 2.8 against 1.95).
    const size_t want_q4 = (size_t)c->num_cu * 4 * (size_t)(want_x2 > 0 ? want_x2 : 1) / 2 * 12;
    const bool q4 = v->q4 && !lane_kernel && (q4_always || chunks >= want_q4 * 8);
    q4_kernel = q4;
    v->q4_call = q4;
    const size_t want = q4 ? want_q4 : (size_t)c->num_cu * 4 * (size_t)(want_x2 > 0 ? want_x2 : 3) / 2;
    const unsigned tl_min = q4 ? 4u : 1u;
    if (forced > 0) TL = (unsigned)forced;
    else while (TL > tl_min && TL % 2 == 0 && chunks / TL < want) TL /= 2;
  }
  std::vector<vit_job> jobs;
  const int cur = v->current_sync;
  {
    unsigned long long cstart = 0;
    size_t ri = 0;                          // first resync chunk beyond cstart (both ascend)
    // first tile ends at the first resync chunk (or after TL chunks)
    while (cstart < chunks) {
      while (ri < rs.size() && rs[ri] <= cstart) ++ri;
      const unsigned long long next_rs = ri < rs.size() ? rs[ri] : chunks;
      unsigned long long cend = cstart + TL;
      if (cend > next_rs) cend = next_rs;
      if (cend > chunks) cend = chunks;
      vit_job j;
      j.first_chunk = cstart; j.n_chunks = (unsigned)(cend - cstart);
      j.warm = cstart == 0 ? 0u : (unsigned)(cstart < (unsigned long long)kWarm ? cstart : (unsigned long long)kWarm);
      j.sync = cur; j.from_state = cstart == 0 ? cur : -1; j.emit = 1; j.chunk_step = 1;
      // tiles that cannot warm up fully (near the start) extend tile 0 instead
      if (cstart != 0 && j.warm < (unsigned)kWarm) {
        jobs.back().n_chunks += j.n_chunks;
      } else jobs.push_back(j);
      cstart = cend;
    }
  }
  const size_t n_main = jobs.size();
  // The other alignments decode only the resync chunks (stride P), each from its own carried state.  Their "virtual
  // stream" is tiled and verified like the main one; their tiles ride in the SAME launch as the main tiles (nothing in
  // them depends on the main alignment's results), after the main slots.
  const bool have_others = !rs.empty() && v->nsyncs > 1;
  // tile length of the other alignments (in resync chunks): the longest of 32, 16, 8 that still leaves a thousand tiles (a
  // fifth of the work is warm-up at 32, half at 8, two thirds at 4)
  unsigned TLo = 4;
  for (unsigned t = 32; t >= 8; t /= 2)
    if ((size_t)(v->nsyncs - 1) * ((rs.size() + t - 1) / t) >= 1024) { TLo = t; break; }
  // warm-up of the other alignments' tiles: they decode a wrong alignment (noise-like input), whose survivors merge more
  // slowly — with the main tiles' 4 chunks a third of their seams needed a fix-up round, with 8 about one in thirty on QPSK
  // (none on 8PSK), with 12 none in the bench's streams.  A fix-up round is a second launch → readback on the critical path
  // of the call whatever the number of tiles in it (c3: 9.0 ms per call with it, 5.5–6 without), a longer warm-up is a few per
  // cent more trellis steps: the decoder starts at 8 and adds 4 (up to 16) whenever a call had to fix up seams of the other
  // alignments.  (The output never depends on it.)
  const int wo_env = getenv("LSDR_VIT_WO") ? atoi(getenv("LSDR_VIT_WO")) : 0;   // tuning hook
  const unsigned Wo = wo_env > 0 ? (unsigned)wo_env : v->warm_others < (unsigned)kWarm ? (unsigned)kWarm : v->warm_others;
  // k_viterbi_q4 runs about one wavefront per SIMD and all of them at once: a call takes as long as its longest wavefront,
  // so the other alignments' tiles (their longer warm-up, the quality of every step on top) are kept no longer than the main ones
  if (q4_kernel) while (TLo > 4 && (TLo + Wo) * 23 > (TL + (unsigned)kWarm) * 22) TLo /= 2;
  const unsigned nrs = (unsigned)rs.size();
  struct other_jobs { std::vector<vit_job> oj; std::vector<int> which, tile_first; unsigned ostride; };   // tile_first: index into rs
  auto build_others = [&](bool sequential, const std::vector<int> &only) {
    other_jobs J;
    for (int s : only) {
      if (sequential) {
        vit_job j;
        j.first_chunk = rs[0]; j.n_chunks = nrs; j.warm = 0; j.sync = s; j.from_state = s; j.emit = 0; j.chunk_step = (unsigned)P;
        J.oj.push_back(j); J.which.push_back(s); J.tile_first.push_back(0);
      } else {
        unsigned r0 = 0;
        while (r0 < nrs) {
          unsigned r1 = r0 + TLo;
          if (r0 == 0 && r1 < Wo + TLo) r1 = Wo + TLo;   // tile 0 is long enough for tile 1 to warm up fully
          if (r1 > nrs) r1 = nrs;
          vit_job j;
          j.first_chunk = rs[r0]; j.n_chunks = r1 - r0; j.warm = r0 == 0 ? 0u : Wo; j.sync = s;
          j.from_state = r0 == 0 ? s : -1; j.emit = 0; j.chunk_step = (unsigned)P;
          J.oj.push_back(j); J.which.push_back(s); J.tile_first.push_back((int)r0);
          r0 = r1;
        }
      }
    }
    J.ostride = 1;
    for (auto &j : J.oj) if (j.n_chunks > J.ostride) J.ostride = j.n_chunks;
    return J;
  };
  std::vector<int> all_others;
  if (have_others) for (int s = 0; s < v->nsyncs; ++s) if (s != cur) all_others.push_back(s);
  other_jobs first_others = build_others(false, all_others);
  const size_t n_first = first_others.oj.size();
  unsigned stride = 1;
  for (auto &j : jobs) if (j.n_chunks > stride) stride = j.n_chunks;
  if (n_first && first_others.ostride > stride) stride = first_others.ostride;
  std::vector<vit_job> launch_jobs(jobs);
  launch_jobs.insert(launch_jobs.end(), first_others.oj.begin(), first_others.oj.end());
  vit_args main_args;
  int rc = vit_launch(v, in, out, launch_jobs, stride, false, phase0, nullptr, nullptr, false, 0, true, n_main, &main_args);
  if (rc) return rc;
  // ---- seam check and fix-up rounds, main and other alignments together.  A tile whose speculative start state differs
  // from its predecessor's end state is decoded again from that end state (read on the device from the slot array; results
  // land in the tile's own slots); a changed end state shows up at the next seam in the next round.  The seam check is
  // redone over all tiles after every round, and it compares the state each job actually started from, so the result is
  // exact whatever the interleaving.  Only the flags and the per-chunk totals come back to the host.
  const size_t n_total = n_main + n_first;
  std::vector<unsigned char> seam(n_total, 0);          // slot k continues slot k−1
  for (size_t k = 1; k < n_main; ++k) seam[k] = 1;
  for (size_t k = 0; k < n_first; ++k) seam[n_main + k] = first_others.tile_first[k] != 0;
  std::vector<int> bad(n_total, 0), totals_all(n_total * stride);
  vit_state main_end;
  std::vector<vit_state> first_end(v->nsyncs);
  v->last_tiles = (unsigned)n_main; v->last_bad = 0;
  vt.tiles = (unsigned)n_total;
  // The FIRST repair round runs on the device, behind the main launch, without the host: seam check into a second flag array, then the
  // launch's own job list once more on the lane = state kernel with those flags as a predicate (vit_args::cond) — a wavefront whose seam
  // holds leaves at once, one whose seam failed decodes its tile again from the end state of the slot before it.  What the host then
  // reads is the state after that round: a call whose failed seams settle in one round (all of them, in the bench's streams) has ONE
  // launch → readback, like a call without any.  The flags the round acted on come back with it (statistics, warm-up adaptation).
  // INVARIANTS this round relies on (advisor, round 5):
  //  * one job per slot, in slot order: the grid below is sized from n_total, the kernel indexes main_args.jobs — so the launch's job
  //    list must have exactly n_total entries (checked here, not assumed);
  //  * ADJACENT failed seams run concurrently: job k reads d_end[k−1] while job k−1 may be rewriting it.  That is allowed because nothing
  //    is trusted afterwards — every job records in d_begin[k] the state it ACTUALLY started from, the host's round 0 below re-runs
  //    k_vit_verify over all seams (d_begin[k] against the FINAL d_end[k−1]) and decodes again, through the host, whatever still
  //    disagrees; so a torn or stale read costs one more round, never a wrong byte.  `dev_repaired` therefore counts the tiles the device
  //    round DECODED AGAIN, not the tiles it settled: what still needed the host shows in `host_rounds` (lsdr_viterbi_repair_stats).
  if (launch_jobs.size() != n_total) { lsdr_set_error("viterbi_sync: %zu jobs for %zu slots", launch_jobs.size(), n_total); return LSDR_E_ARG; }
  const bool dev_repair = n_total > 1 && getenv("LSDR_VIT_HOST_REPAIR") == nullptr;   // (test hook, read per call: every round through the host)
  int *const d_bad0 = v->d_bad + v->jobs_cap;
  std::vector<int> bad0;
  if (dev_repair) {
    LSDR_HIP(hipMemsetAsync(d_bad0, 0, n_total * sizeof(int), c->stream));
    hipLaunchKernelGGL(k_vit_verify, dim3((unsigned)(n_total - 1)), dim3(64), 0, c->stream,
                       (const vit_state *)v->d_begin, (const vit_state *)v->d_end, (unsigned)n_total, d_bad0);
    vit_args fa = main_args;
    fa.cond = d_bad0; fa.states_in = v->d_end; fa.chunk_states = nullptr;
    const dim3 grid((unsigned)((n_total + kVitWaves - 1) / kVitWaves)), block(kVitWaves * 64);
    const bool generic_only = getenv("LSDR_VIT_GENERIC") != nullptr;
    if (fa.C.nus == 2 && fa.C.bits_out == 2 && !generic_only) hipLaunchKernelGGL(k_viterbi<2>, grid, block, 0, c->stream, fa);
    else if (fa.C.nus == 4 && fa.C.bits_out == 3 && !generic_only) hipLaunchKernelGGL(k_viterbi<4>, grid, block, 0, c->stream, fa);
    else hipLaunchKernelGGL(k_viterbi<0>, grid, block, 0, c->stream, fa);
    LSDR_HIP(hipGetLastError());
    bad0.resize(n_total);
  }
  for (int round = 0;; ++round) {
    LSDR_HIP(hipMemsetAsync(v->d_bad, 0, n_total * sizeof(int), c->stream));
    if (n_total > 1) hipLaunchKernelGGL(k_vit_verify, dim3((unsigned)(n_total - 1)), dim3(64), 0, c->stream,
                                        (const vit_state *)v->d_begin, (const vit_state *)v->d_end, (unsigned)n_total, v->d_bad);
    LSDR_TRY(lsdr_stage_d2h(c, bad.data(), v->d_bad, n_total * sizeof(int)));
    if (dev_repair && round == 0) LSDR_TRY(lsdr_stage_d2h(c, bad0.data(), d_bad0, n_total * sizeof(int)));
    LSDR_TRY(lsdr_stage_d2h(c, totals_all.data(), v->d_totals, n_total * stride * sizeof(int)));
    // end states the host needs if this round turns out to be the last (later launches reuse the slots): the main
    // alignment's last tile, each other alignment's last tile
    LSDR_TRY(lsdr_stage_d2h(c, &main_end, v->d_end + (n_main - 1), sizeof(vit_state)));
    for (size_t k = 0; k < n_first; ++k)
      if (k + 1 == n_first || first_others.which[k + 1] != first_others.which[k])
        LSDR_TRY(lsdr_stage_d2h(c, &first_end[first_others.which[k]], v->d_end + (n_main + k), sizeof(vit_state)));
    VIT_SYNC(c, vt);
    for (size_t k = 0; k < n_total; ++k) if (!seam[k]) bad[k] = 0;
    if (dev_repair && round == 0) {      // what the device-side round re-decoded
      unsigned nrep = 0;
      bool others = false;
      for (size_t k = 1; k < n_total; ++k)
        if (seam[k] && bad0[k]) { ++nrep; if (k >= n_main) others = true; }
      v->last_bad += nrep; v->dev_repaired += nrep; vt.fixups += nrep;
      if (others && v->warm_others < 16) v->warm_others += 4;
    }
    if (round >= 6) break;
    std::vector<vit_job> fj;
    for (size_t k = 1; k < n_total; ++k)
      if (bad[k]) {
        vit_job j = launch_jobs[k];
        j.warm = 0; j.from_state = (int)(k - 1); j.slot = (unsigned)k;
        fj.push_back(j);
      }
    if (fj.empty()) break;
    ++v->host_rounds;
    if (round == 0 && !dev_repair && v->warm_others < 16)
      for (const vit_job &j : fj) if (j.sync != cur) { v->warm_others += 4; break; }
    v->last_bad += (unsigned)fj.size();
    vt.fixups += (unsigned)fj.size();
    if (vt.on && round == 0) {       // LSDR_VIT_TIMING: which seams failed (slot, alignment, first chunk, chunks)
      fprintf(stderr, "viterbi fix-ups (n_main %zu, n_total %zu):", n_main, n_total);
      for (size_t q = 0; q < fj.size() && q < 12; ++q) fprintf(stderr, " [slot %u sync %d first %llu n %u]", fj[q].slot, fj[q].sync, (unsigned long long)fj[q].first_chunk, fj[q].n_chunks);
      fprintf(stderr, "\n");
    }
    rc = vit_launch(v, in, out, fj, stride, false, phase0, nullptr, v->d_end, true, n_total);
    if (rc) return rc;
  }
  std::vector<int> totals_main(totals_all.begin(), totals_all.begin() + n_main * stride);
  // ---- last resort (seams that keep failing): everything from the first bad seam on, sequentially, from the previous
  // tile's end state (exact by construction).
  size_t first_bad = n_main;
  for (size_t j = 1; j < n_main; ++j) if (bad[j]) { first_bad = j; break; }
  std::vector<vit_state> main_first;           // state after the first chunk of each tile (alignment switches); sized only on the sequential path
  bool main_first_on_device = false;
  if (first_bad < n_main) {
    for (size_t j = first_bad; j < n_main; ++j) v->last_bad += bad[j] ? 1u : 0u;
    // keep results of tiles < first_bad; redo the rest as ONE sequential job starting from end_states[first_bad-1]
    std::vector<vit_state> ends(n_main);
    LSDR_TRY(lsdr_stage_d2h(c, ends.data(), v->d_end, n_main * sizeof(vit_state)));
    main_first.resize(n_main);
    LSDR_TRY(lsdr_stage_d2h(c, main_first.data(), v->d_first, n_main * sizeof(vit_state)));
    VIT_SYNC(c, vt);
    std::vector<vit_state> keep = v->states;
    v->states[cur] = ends[first_bad - 1];
    vit_job j;
    j.first_chunk = jobs[first_bad].first_chunk; j.n_chunks = (unsigned)(chunks - j.first_chunk);
    j.warm = 0; j.sync = cur; j.from_state = cur; j.emit = 1; j.chunk_step = 1;
    std::vector<vit_job> redo(1, j);
    rc = vit_launch(v, in, out, redo, j.n_chunks, true, phase0);
    if (rc) return rc;
    std::vector<int> tot(j.n_chunks);
    std::vector<vit_state> cst(j.n_chunks);
    LSDR_TRY(lsdr_stage_d2h(c, tot.data(), v->d_totals, j.n_chunks * sizeof(int)));
    LSDR_TRY(lsdr_stage_d2h(c, cst.data(), v->d_chunk, j.n_chunks * sizeof(vit_state)));
    LSDR_TRY(lsdr_stage_d2h(c, &main_end, v->d_end, sizeof(vit_state)));
    VIT_SYNC(c, vt);
    v->states = keep;
    // splice the sequential results back into the per-tile bookkeeping
    for (size_t t = first_bad; t < n_main; ++t) {
      const unsigned off = (unsigned)(jobs[t].first_chunk - j.first_chunk);
      for (unsigned q = 0; q < jobs[t].n_chunks; ++q) totals_main[t * stride + q] = tot[off + q];
      main_first[t] = cst[off];
    }
  } else {
    main_first_on_device = true;   // fetched per tile only if an alignment switch needs it (2.8 MB per call otherwise)
  }

  // ---- other alignments: sequential over the resync chunks, exact carried state
  size_t used_chunks = chunks;
  int new_sync = cur;
  std::vector<std::vector<vit_state> > other_states(v->nsyncs);
  std::vector<std::vector<int> > other_totals(v->nsyncs);
  std::vector<vit_state> other_end(v->nsyncs);
  std::vector<int> other_ok(v->nsyncs, 1);
  if (have_others) {
    // A decoder whose seams do not verify (wrong alignments see noise-like input, survivors may merge slowly) is redone
    // sequentially.  `finish_others` takes the downloaded results of a launch of J's jobs (per-chunk totals with row
    // stride `tstride`, begin/end states per job, per-chunk states of a sequential launch).
    auto finish_others = [&](bool sequential, const other_jobs &J, const std::vector<int> &tot, unsigned tstride,
                             const std::vector<vit_state> &hb, std::vector<vit_state> he, const std::vector<vit_state> &cst) -> int {
      const std::vector<vit_job> &oj = J.oj;
      const std::vector<int> &which = J.which, &tile_first = J.tile_first;
      for (size_t k = 0; k < oj.size(); ++k) {
        const int s = which[k];
        if (tile_first[k] == 0) { other_totals[s].assign(nrs, 0); other_states[s].clear(); other_ok[s] = 1; }
        for (unsigned q = 0; q < oj[k].n_chunks; ++q) other_totals[s][tile_first[k] + q] = tot[k * tstride + q];
        if (sequential) other_states[s].assign(cst.begin() + k * tstride, cst.begin() + k * tstride + nrs);
      }
      if (!sequential) {
        // fix-up rounds: a tile whose speculative start state differs from its predecessor's true end state is
        // re-decoded from that end state; a changed end state propagates to the next seam in the next round.
        std::vector<vit_state> start_used = hb;
        for (int round = 0;; ++round) {
          std::vector<size_t> badk;
          for (size_t k = 0; k < oj.size(); ++k)
            if (tile_first[k] != 0 && memcmp(&start_used[k], &he[k - 1], sizeof(vit_state)) != 0) badk.push_back(k);
          if (badk.empty()) break;
          v->last_bad += (unsigned)badk.size();
          if (round >= 6) { for (size_t k : badk) other_ok[which[k]] = 0; break; }   // pathological: sequential fallback
          std::vector<vit_job> fj;
          std::vector<vit_state> starts;
          unsigned fstride = 1;
          for (size_t k : badk) {
            vit_job j = oj[k];
            j.warm = 0; j.from_state = (int)starts.size();
            starts.push_back(he[k - 1]);
            fj.push_back(j);
            if (j.n_chunks > fstride) fstride = j.n_chunks;
          }
          int rc2 = vit_launch(v, in, out, fj, fstride, false, phase0, &starts, nullptr, false, 0, false);
          if (rc2) return rc2;
          std::vector<int> ftot(fj.size() * fstride);
          std::vector<vit_state> fe(fj.size());
          LSDR_TRY(lsdr_stage_d2h(c, ftot.data(), v->d_totals, ftot.size() * sizeof(int)));
          LSDR_TRY(lsdr_stage_d2h(c, fe.data(), v->d_end, fj.size() * sizeof(vit_state)));
          VIT_SYNC(c, vt);
          for (size_t i = 0; i < badk.size(); ++i) {
            const size_t k = badk[i];
            start_used[k] = starts[i];
            he[k] = fe[i];
            for (unsigned q = 0; q < oj[k].n_chunks; ++q) other_totals[which[k]][tile_first[k] + q] = ftot[i * fstride + q];
          }
        }
      }
      for (size_t k = 0; k < oj.size(); ++k) other_end[which[k]] = he[k];
      return LSDR_OK;
    };
    auto run_others = [&](bool sequential, const std::vector<int> &only) -> int {
      const other_jobs J = build_others(sequential, only);
      int rc2 = vit_launch(v, in, out, J.oj, J.ostride, sequential, phase0, nullptr, nullptr, false, 0, false);
      if (rc2) return rc2;
      std::vector<int> tot(J.oj.size() * J.ostride);
      std::vector<vit_state> hb(J.oj.size()), he(J.oj.size()), cst;
      LSDR_TRY(lsdr_stage_d2h(c, tot.data(), v->d_totals, tot.size() * sizeof(int)));
      LSDR_TRY(lsdr_stage_d2h(c, hb.data(), v->d_begin, J.oj.size() * sizeof(vit_state)));
      LSDR_TRY(lsdr_stage_d2h(c, he.data(), v->d_end, J.oj.size() * sizeof(vit_state)));
      if (sequential) {
        cst.resize(J.oj.size() * J.ostride);
        LSDR_TRY(lsdr_stage_d2h(c, cst.data(), v->d_chunk, cst.size() * sizeof(vit_state)));
      }
      VIT_SYNC(c, vt);
      return finish_others(sequential, J, tot, J.ostride, hb, he, cst);
    };
    // the tiles that rode in the main launch are verified already; an alignment with a seam that never settled goes sequential
    for (int sidx : all_others) { other_totals[sidx].assign(nrs, 0); other_states[sidx].clear(); other_ok[sidx] = 1; other_end[sidx] = first_end[sidx]; }
    for (size_t k = 0; k < n_first; ++k) {
      const int sidx = first_others.which[k];
      for (unsigned q = 0; q < first_others.oj[k].n_chunks; ++q)
        other_totals[sidx][first_others.tile_first[k] + q] = totals_all[(n_main + k) * stride + q];
      if (bad[n_main + k]) other_ok[sidx] = 0;
    }
    {
      std::vector<int> redo;
      for (int s : all_others) if (!other_ok[s]) redo.push_back(s);
      if (!redo.empty()) { rc = run_others(true, redo); if (rc) return rc; }
    }
    std::vector<vit_state> st = v->states;
    size_t tj = 0;                          // tile holding resync chunk r: tiles and resync chunks both ascend
    for (size_t r = 0; r < rs.size(); ++r) {
      // alignment decision after this resync chunk (dvb.h:1401-1410): s ascending from best = current, strict '>'
      while (tj + 1 < n_main && jobs[tj + 1].first_chunk <= rs[r]) ++tj;
      const int tcur = totals_main[tj * stride + (unsigned)(rs[r] - jobs[tj].first_chunk)];
      int best = cur, bt = tcur;
      for (int s = 0; s < v->nsyncs; ++s) {
        const int ts = s == cur ? tcur : other_totals[s][r];
        if (ts > bt) { best = s; bt = ts; }
      }
      static const bool vit_debug = getenv("LSDR_VIT_DEBUG") != nullptr;
      if (vit_debug) {
        fprintf(stderr, "VIT out=%zu cur=%d best=%d :", (size_t)(v->dbg_chunks + rs[r] + 1) * bytes_per_chunk, cur, best);
        for (int s = 0; s < v->nsyncs; ++s) fprintf(stderr, " %d", s == cur ? tcur : other_totals[s][r]);
        fprintf(stderr, "\n");
      }
      if (best != cur) {
        // switch: everything after this chunk must be decoded with the new alignment → stop here
        used_chunks = rs[r] + 1;
        new_sync = best;
        // carried states: other alignments as of this chunk (sequential re-run records them); the old current one
        // as of the end of this chunk
        {
          std::vector<int> need;
          for (int s2 : all_others) if (other_states[s2].size() != nrs) need.push_back(s2);
          if (!need.empty()) { rc = run_others(true, need); if (rc) return rc; }
          for (int s2 : all_others) st[s2] = other_states[s2][r];
        }
        vit_state old_cur;
        if (rs[r] == jobs[tj].first_chunk) {
          if (main_first_on_device) LSDR_HIP(hipMemcpy(&old_cur, v->d_first + tj, sizeof(vit_state), hipMemcpyDeviceToHost));
          else old_cur = main_first[tj];
        }
        else {   // resync chunk inside tile 0 (its first chunk is not a resync chunk): recompute sequentially
          if (tj != 0) { lsdr_set_error("viterbi_sync: internal tiling error"); return LSDR_E_ARG; }
          vit_job j;
          j.first_chunk = 0; j.n_chunks = (unsigned)(rs[r] + 1);
          j.warm = 0; j.sync = cur; j.from_state = cur; j.emit = 1; j.chunk_step = 1;
          std::vector<vit_job> one(1, j);
          rc = vit_launch(v, in, out, one, j.n_chunks, false, phase0, nullptr, nullptr, false, 0, false);
          if (rc) return rc;
          LSDR_HIP(hipMemcpy(&old_cur, v->d_end, sizeof(vit_state), hipMemcpyDeviceToHost));
        }
        for (int s = 0; s < v->nsyncs; ++s) v->states[s] = s == cur ? old_cur : st[s];
        break;
      }
    }
    if (new_sync == cur) {
      for (int s = 0; s < v->nsyncs; ++s) if (s != cur) v->states[s] = other_end[s];
      v->states[cur] = main_end;
    }
  } else {
    v->states[cur] = main_end;
  }
  if (new_sync != v->current_sync) v->budget_chunks = (size_t)v->resync_period;
  else if (v->budget_chunks < ((size_t)1 << 40)) v->budget_chunks *= 2;
  v->current_sync = new_sync;
  v->resync_phase = (int)(((unsigned long long)phase0 + used_chunks) % (unsigned)P);
  v->dbg_chunks += used_chunks;
  *consumed = used_chunks * sym_per_chunk;
  *produced = used_chunks * bytes_per_chunk;
  return LSDR_OK;
}

}  // extern "C"
