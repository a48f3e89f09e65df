/* oracle/lsdr_oracle_fec.c — CPU ORACLE (test infrastructure).
 * DVB-S FEC tail: deconvol_sync, viterbi_sync (+ viterbi_dec/trellis/bitpath), mpeg_sync,
 * deinterleaver, RS(204,188) decoder, derandomizer — restated from dvb.h, viterbi.h, rs.h.
 * All integer arithmetic; every function follows one reference run() call over one buffer. */
#include "lsdr_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int par64(uint64_t x) { return __builtin_parityll(x); }
static int log2u(uint64_t x) { int n = -1; for (; x; ++n, x >>= 1); return n; }   /* dvb.h:157-161 */

#define SIZE_RSPACKET 204
#define SIZE_TSPACKET 188
#define MPEG_SYNC 0x47
#define MPEG_SYNC_INV 0xb8
#define MPEG_SYNC_CORRUPTED 0x55
#define DVBS_G1 0171
#define DVBS_G2 0133

/* ======================================================================= deconvol_sync, dvb.h:122-476 */
#define DC_TRACEBACK 64
#define DC_NSYNCS 4
struct lo_deconv {
  uint32_t conv[2], punct[2];
  int punctperiod, punctweight;
  uint64_t response[64], deconv[8], deconv2[8];
  struct dsync { uint8_t lut[2][2]; uint64_t in; int n_in; uint64_t out; int n_out; uint64_t in2; int n_in2, n_out2; } syncs[DC_NSYNCS];
  int locked, skip, fastlock;
};

/* dvb.h:163-179 */
static uint64_t dc_convolve(const lo_deconv *d, uint64_t s) {
  int sbits = log2u(s) + 1;
  uint64_t iq = 0;
  unsigned char state = 0;
  for (int b = sbits - 1; b >= 0; --b) {
    unsigned char bit = (s >> b) & 1;
    state = (state >> 1) | (bit << 6);
    for (int j = 0; j < 2; ++j) {
      unsigned char xy = par64(state & d->conv[j]);
      if (d->punct[j] & (1 << (b % d->punctperiod))) iq = (iq << 1) | xy;
    }
  }
  return iq;
}

/* dvb.h:205-224 */
static void dc_solve_rec(const lo_deconv *d, uint64_t prefix, int nprefix, uint64_t exp, uint64_t *best) {
  if (prefix > *best) return;
  if (nprefix > 64) return;
  int solved = 1;
  for (int b = 0; b < 64; ++b) {
    if (par64(prefix & d->response[b]) != (int)((exp >> b) & 1)) {
      if (nprefix >= 64 || (d->response[b] >> nprefix) == 0) return;
      solved = 0;
    }
  }
  if (solved) { *best = prefix; return; }
  dc_solve_rec(d, prefix, nprefix + 1, exp, best);
  dc_solve_rec(d, prefix | ((uint64_t)1 << nprefix), nprefix + 1, exp, best);
}

/* alternate polynomials for fastlock, dvb.h:238-266 */
static uint64_t dc_alt(uint64_t d) {
  static const uint64_t tab[][2] = {
      {0x3baULL, 0x38ccaULL},
      {0xf29ULL, 0x3c569329ULL}, {0x3c552ULL, 0x1dee1cULL}, {0x7948ULL, 0x1e2b49948ULL}, {0x1deULL, 0x1e2a90ULL},
      {0xf247ULL, 0xfd6383bULL}, {0xfd9eeULL, 0xfd91392ULL}, {0xf248d8ULL, 0xfd9eef18ULL},
      {0xf5727fULL, 0x3d5c909758fULL}, {0x3d5c90aaULL, 0x0f5727f0229c90aaULL}, {0x3daa371cULL, 0x3d5f45630ecULL},
      {0xf5727ff48ULL, 0xf57d28260348ULL}, {0xf57d28260ULL, 0xf5727ff48128260ULL},
      {0xfbeac76c454fULL, 0xfb11d6ba045a8fULL}, {0xfb11d6baULL, 0xfbea3c7d930e16baULL},
      {0xfb112d5038dcULL, 0xfb112d5038271cULL}, {0xfbea3c7d68ULL, 0xfbeac7975462a8ULL},
      {0xfb112d50ULL, 0xfbea3c86793290ULL}, {0xfb112dabd2e0ULL, 0xfb112d50c3cd20ULL},
      {0xfb11d640ULL, 0xfbea3c8679c980ULL}};
  uint64_t d2 = d;
  for (size_t i = 0; i < sizeof(tab) / sizeof(tab[0]); ++i)
    if (d == tab[i][0]) d2 = tab[i][1];
  return d2;
}

/* make_deconvol_sync_simple, dvb.h:480-513 + ctor dvb.h:124-152 */
lo_deconv *lo_deconv_new(int rate, int fastlock) {
  lo_deconv *d = (lo_deconv *)calloc(1, sizeof(*d));
  uint32_t pX, pY;
  switch (rate) {
    case LO_FEC12: pX = 0x1; pY = 0x1; break;
    case LO_FEC23: case LO_FEC46: pX = 0xa; pY = 0xf; break;
    case LO_FEC34: pX = 0x5; pY = 0x6; break;
    case LO_FEC56: pX = 0x15; pY = 0x1a; break;
    case LO_FEC78: pX = 0x45; pY = 0x7a; break;
    default: pX = pY = 1;
  }
  d->conv[0] = DVBS_G1; d->conv[1] = DVBS_G2;
  d->punct[0] = pX; d->punct[1] = pY;
  d->fastlock = fastlock;
  for (int i = 0; i < 2; ++i) {
    int nbits = log2u(d->punct[i]) + 1;
    if (nbits > d->punctperiod) d->punctperiod = nbits;
    d->punctweight += __builtin_popcount(d->punct[i]);
  }
  /* inverse_convolution, dvb.h:228-292 */
  for (int sbit = 0; sbit < 64; ++sbit) d->response[sbit] = dc_convolve(d, (uint64_t)1 << sbit);
  for (int b = 0; b < d->punctperiod; ++b) {
    d->deconv[b] = ~(uint64_t)0;
    dc_solve_rec(d, 0, 0, (uint64_t)(1 << b), &d->deconv[b]);
    d->deconv2[b] = dc_alt(d->deconv[b]);
  }
  /* init_syncs, dvb.h:309-366 */
  for (int id = 0; id < DC_NSYNCS; ++id)
    for (int re_pos = 0; re_pos <= 1; ++re_pos)
      for (int im_pos = 0; im_pos <= 1; ++im_pos) {
        int re_neg = !re_pos, I = 0, Q = 0;
        switch (id) {
          case 0: I = re_pos ? 0 : 1; Q = im_pos ? 0 : 1; break;
          case 1: I = im_pos ? 0 : 1; Q = re_neg ? 0 : 1; break;
          case 2: I = re_pos ? 0 : 1; Q = im_pos ? 1 : 0; break;
          case 3: I = im_pos ? 1 : 0; Q = re_neg ? 0 : 1; break;
        }
        d->syncs[id].lut[re_pos][im_pos] = (I << 1) | Q;
      }
  d->locked = 0;
  return d;
}
void lo_deconv_free(lo_deconv *d) { free(d); }
void lo_deconv_info(const lo_deconv *d, uint64_t *deconv, uint64_t *deconv2, int *period, int *weight) {
  for (int b = 0; b < d->punctperiod; ++b) { deconv[b] = d->deconv[b]; deconv2[b] = d->deconv2[b]; }
  *period = d->punctperiod; *weight = d->punctweight;
}
/* dvb.h:185-193 */
void lo_deconv_next_sync(lo_deconv *d) {
  ++d->locked;
  if (d->locked == DC_NSYNCS) { d->locked = 0; d->skip = 1; }
}
int lo_deconv_locked(const lo_deconv *d) { return d->locked; }

/* dvb.h:375-394 */
static uint8_t dc_readbyte(lo_deconv *d, struct dsync *s, const lo_softsymbol **pp) {
  const lo_softsymbol *p = *pp;
  while (s->n_out < 8) {
    uint64_t iq = s->in;
    while (s->n_in < DC_TRACEBACK) {
      uint8_t iqbits = s->lut[(p->symbol & 2) ? 1 : 0][p->symbol & 1];
      ++p;
      iq = (iq << 2) | iqbits;
      s->n_in += 2;
    }
    s->in = iq;
    for (int b = d->punctperiod - 1; b >= 0; --b) s->out = (s->out << 1) | (uint64_t)par64(iq & d->deconv[b]);
    s->n_out += d->punctperiod;
    s->n_in -= d->punctweight;
  }
  uint8_t res = (s->out >> (s->n_out - 8)) & 255;
  s->n_out -= 8;
  *pp = p;
  return res;
}
/* dvb.h:396-417 */
static unsigned long dc_readerrors(lo_deconv *d, struct dsync *s, const lo_softsymbol **pp) {
  const lo_softsymbol *p = *pp;
  unsigned long res = 0;
  while (s->n_out2 < 8) {
    uint64_t iq = s->in2;
    while (s->n_in2 < DC_TRACEBACK) {
      uint8_t iqbits = s->lut[(p->symbol & 2) ? 1 : 0][p->symbol & 1];
      ++p;
      iq = (iq << 2) | iqbits;
      s->n_in2 += 2;
    }
    s->in2 = iq;
    for (int b = d->punctperiod - 1; b >= 0; --b)
      if (par64(iq & d->deconv2[b]) != par64(iq & d->deconv[b])) ++res;
    s->n_out2 += d->punctperiod;
    s->n_in2 -= d->punctweight;
  }
  s->n_out2 -= 8;
  *pp = p;
  return res;
}
/* run_decoding, dvb.h:419-470: one run() call */
size_t lo_deconv_run(lo_deconv *d, const lo_softsymbol *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed) {
  size_t pos = d->skip;   /* in.read(skip) */
  d->skip = 0;
  *consumed = pos;
  if (n_in < pos) { *consumed = n_in; return 0; }
  size_t readable = n_in - pos;
  if (readable < 64) return 0;
  int maxrd = (int)((readable - 64) / (d->punctweight / 2) * d->punctperiod / 8);
  int maxwr = (int)cap;
  int n = maxrd < maxwr ? maxrd : maxwr;
  if (!n) return 0;
  if (n < 32) return 0;
  if (d->fastlock) {
    unsigned long errors_best = 1 << 30;
    int best = 0;
    for (int s = 0; s < DC_NSYNCS; ++s) {
      const lo_softsymbol *pin = in + pos;
      unsigned long errors = 0;
      for (int c = n; c--;) errors += dc_readerrors(d, &d->syncs[s], &pin);
      if (errors < errors_best) { errors_best = errors; best = s; }
    }
    if (best != d->locked) d->locked = best;
    if (errors_best > (unsigned long)(n * 8 / 3)) d->skip = 1;
  }
  const lo_softsymbol *pin = in + pos;
  for (int k = 0; k < n; ++k) out[k] = dc_readbyte(d, &d->syncs[d->locked], &pin);
  *consumed = pin - in;
  return n;
}

/* ======================================================================= viterbi, viterbi.h + dvb.h:1173-1416 */
typedef struct { int bits_in, bits_out, nus, ncs, nbits, depth, pathbits; const uint16_t *polys; } vspec;
static const uint16_t polys12[] = {DVBS_G1, DVBS_G2};
static const uint16_t polys23[] = {DVBS_G1, DVBS_G2, DVBS_G2 << 1};
static const uint16_t polys46[] = {DVBS_G1, DVBS_G2, DVBS_G2 << 1, DVBS_G1 << 2, DVBS_G2 << 2, DVBS_G2 << 3};
static const uint16_t polys34[] = {DVBS_G1, DVBS_G2, DVBS_G2 << 1, DVBS_G1 << 2};
static const uint16_t polys45[] = {DVBS_G1, DVBS_G2, DVBS_G2 << 1, DVBS_G1 << 2, DVBS_G1 << 3};
static const uint16_t polys56[] = {DVBS_G1, DVBS_G2, DVBS_G2 << 1, DVBS_G1 << 2, DVBS_G2 << 3, DVBS_G1 << 4};
static const uint16_t polys78[] = {DVBS_G1, DVBS_G2, DVBS_G2 << 1, DVBS_G2 << 2, DVBS_G2 << 3, DVBS_G1 << 4, DVBS_G2 << 5, DVBS_G1 << 6};
/* fec_specs (dvb.h:556-566) and the path/trellis typedefs (dvb.h:1179-1212) */
static int vspec_for(int rate, vspec *v) {
  switch (rate) {
    case LO_FEC12: *v = (vspec){1, 2, 2, 4, 1, 32, 32, polys12}; return 0;
    case LO_FEC23: *v = (vspec){2, 3, 4, 8, 3, 21, 64, polys23}; return 0;
    case LO_FEC46: *v = (vspec){4, 6, 16, 64, 4, 16, 64, polys46}; return 0;
    case LO_FEC34: *v = (vspec){3, 4, 8, 16, 3, 21, 64, polys34}; return 0;
    case LO_FEC45: *v = (vspec){4, 5, 16, 32, 4, 16, 64, polys45}; return 0;
    case LO_FEC56: *v = (vspec){5, 6, 32, 64, 5, 12, 64, polys56}; return 0;
    case LO_FEC78: *v = (vspec){7, 8, 128, 256, 7, 9, 64, polys78}; return 0;
  }
  return -1;
}
#define VS_NSTATES 64
#define VS_NOSTATE 65
typedef struct { uint8_t pred[VS_NSTATES][256], us[VS_NSTATES][256]; } vtrellis;
/* trellis::init_convolutional, viterbi.h:59-92 */
static void vtrellis_init(vtrellis *t, const vspec *v) {
  memset(t->pred, VS_NOSTATE, sizeof(t->pred));
  int nG = log2u(v->ncs);
  for (int s = 0; s < VS_NSTATES; ++s)
    for (int us = 0; us < v->nus; ++us) {
      uint64_t shiftreg = s;
      int us_rev = 0;
      for (int b = 1; b < v->nus; b *= 2) if (us & b) us_rev |= (v->nus / 2 / b);
      shiftreg |= (uint64_t)us_rev * VS_NSTATES;
      uint32_t cs = 0;
      for (int g = 0; g < nG; ++g) cs = (cs << 1) | par64(shiftreg & v->polys[g]);
      shiftreg /= v->nus;
      t->pred[shiftreg][cs] = s;
      t->us[shiftreg][cs] = us;
    }
}
typedef struct { int32_t cost[2][VS_NSTATES]; uint64_t path[2][VS_NSTATES]; int bank; } vdec;
/* viterbi_dec::update(TCS cs, TBM cost, TPM *quality) = update(1,&cs,&cost,q), viterbi.h:202-267 */
static int vdec_update(vdec *d, const vtrellis *t, const vspec *v, int cs1, int32_t cost1, int32_t *quality) {
  const int32_t max_tpm = 0x7fffffff;
  int32_t best_tpm = max_tpm, best2_tpm = max_tpm;
  int best_state = 0;
  const int32_t *oc = d->cost[d->bank]; const uint64_t *op = d->path[d->bank];
  int32_t *nc = d->cost[d->bank ^ 1]; uint64_t *np = d->path[d->bank ^ 1];
  const uint64_t pmask = v->pathbits == 64 ? ~(uint64_t)0 : (((uint64_t)1 << v->pathbits) - 1);
  for (int s = 0; s < VS_NSTATES; ++s) {
    int32_t best_m = max_tpm;
    int bp = -1, bu = 0;
    if (t->pred[s][cs1] != VS_NOSTATE) {
      int32_t m = oc[t->pred[s][cs1]] + cost1;
      if (m <= best_m) { best_m = m; bp = t->pred[s][cs1]; bu = t->us[s][cs1]; }
    }
    if (1 != v->ncs)
      for (int cs = 0; cs < v->ncs; ++cs) {
        if (t->pred[s][cs] == VS_NOSTATE) continue;
        int32_t m = oc[t->pred[s][cs]];
        if (m <= best_m) { best_m = m; bp = t->pred[s][cs]; bu = t->us[s][cs]; }
      }
    np[s] = ((op[bp] << v->nbits) | (uint64_t)bu) & pmask;   /* bitpath::append, viterbi.h:288 */
    nc[s] = best_m;
    if (best_m < best_tpm) { best_state = s; best2_tpm = best_tpm; best_tpm = best_m; }
    else if (best_m < best2_tpm) best2_tpm = best_m;
  }
  d->bank ^= 1;
  for (int s = 0; s < VS_NSTATES; ++s) d->cost[d->bank][s] -= best_tpm;
  if (quality) *quality = best2_tpm - best_tpm;
  return (int)((d->path[d->bank][best_state] >> ((v->depth - 1) * v->nbits)) & ((1 << v->nbits) - 1));   /* bitpath::read */
}

struct lo_viterbi {
  vspec v; vtrellis trell;
  int bits_per_symbol, nsyncs, nshifts, current_sync, resync_phase, resync_period;
  struct { int shift; vdec dec; uint8_t map[256]; } *syncs;
};
/* viterbi_sync ctor, dvb.h:1234-1331; init_map dvb.h:1336-1351 */
lo_viterbi *lo_viterbi_new(int cstln, int rate) {
  lo_viterbi *s = (lo_viterbi *)calloc(1, sizeof(*s));
  if (vspec_for(rate, &s->v)) { free(s); return NULL; }
  lo_cstln_lut *c = (lo_cstln_lut *)malloc(sizeof(*c));
  lo_make_dvbs2_constellation(c, cstln, rate);
  s->bits_per_symbol = log2u(c->nsymbols);
  if (s->bits_per_symbol * (s->v.bits_out / s->bits_per_symbol) != s->v.bits_out) { free(c); free(s); return NULL; }
  int nconj = c->nsymbols == 2 ? 1 : 2;
  int nrot = (c->nsymbols == 2 || c->nsymbols == 4) ? c->nrotations / 2 : c->nrotations;
  s->nshifts = s->v.bits_out / s->bits_per_symbol;
  s->nsyncs = nconj * nrot * s->nshifts;
  s->syncs = calloc(s->nsyncs, sizeof(*s->syncs));
  s->resync_period = 32;
  vtrellis_init(&s->trell, &s->v);
  for (int k = 0; k < s->nsyncs; ++k) {
    int rot = k % nrot, conj = (k / nrot) % nconj, shift = k / nrot / nconj;
    s->syncs[k].shift = shift;
    float angle = 2 * M_PI * rot / c->nrotations;
    float ca = cosf(angle), sa = sinf(angle);
    for (int i = 0; i < c->nsymbols; ++i) {
      int8_t I = c->symbols[i][0], Q = c->symbols[i][1];
      if (conj) Q = -Q;
      int8_t RI = I * ca - Q * sa;
      int8_t RQ = I * sa + Q * ca;
      s->syncs[k].map[i] = c->symbol[(unsigned)(uint8_t)RI * 256 + (uint8_t)RQ];
    }
  }
  free(c);
  return s;
}
void lo_viterbi_free(lo_viterbi *s) { free(s->syncs); free(s); }
void lo_viterbi_set_resync_period(lo_viterbi *s, int p) { s->resync_period = p; }
int lo_viterbi_current_sync(const lo_viterbi *s) { return s->current_sync; }
int lo_viterbi_nsyncs(const lo_viterbi *s) { return s->nsyncs; }
void lo_viterbi_map(const lo_viterbi *s, int sync, uint8_t *map) { memcpy(map, s->syncs[sync].map, 256); }

static int vs_update_sync(lo_viterbi *s, int k, const lo_softsymbol *pin, int32_t *discr) {   /* dvb.h:1353-1364 */
  pin += s->syncs[k].shift;
  int cs = 0; int32_t cost = 0;
  for (int i = 0; i < s->nshifts; ++i, ++pin) {
    cs = ((cs << s->bits_per_symbol) | s->syncs[k].map[pin->symbol]) & 0xff;   /* TCS is uint8_t */
    cost += pin->cost;
  }
  return vdec_update(&s->syncs[k].dec, &s->trell, &s->v, cs, cost, discr);
}
/* viterbi_sync::run, dvb.h:1366-1414 (all chunks that fit) */
size_t lo_viterbi_run(lo_viterbi *s, const lo_softsymbol *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed) {
  const int chunk = 128;
  int discr_delay = 64 / s->v.bits_in;
  size_t pos = 0, nout_bytes = 0;
  while (n_in - pos >= (size_t)(s->nshifts * chunk + (s->nshifts - 1)) && n_in >= pos &&
         (cap - nout_bytes) * 8 >= (size_t)(s->v.bits_in * chunk)) {
    int32_t total[64];
    for (int k = 0; k < s->nsyncs; ++k) total[k] = 0;
    uint64_t outstream = 0; int nout = 0;
    const lo_softsymbol *pin = in + pos;
    for (int b = 0; b < chunk; ++b, pin += s->nshifts) {
      int32_t discr;
      int result = vs_update_sync(s, s->current_sync, pin, &discr);
      outstream = (outstream << s->v.bits_in) | (uint64_t)result;
      nout += s->v.bits_in;
      if (b >= discr_delay) total[s->current_sync] += discr;
      if (!s->resync_phase)
        for (int k = 0; k < s->nsyncs; ++k) {
          if (k == s->current_sync) continue;
          int32_t dk;
          (void)vs_update_sync(s, k, pin, &dk);
          if (b >= discr_delay) total[k] += dk;
        }
      while (nout >= 8) { out[nout_bytes++] = (uint8_t)(outstream >> (nout - 8)); nout -= 8; }
    }
    pos += (size_t)chunk * s->nshifts;
    if (!s->resync_phase) {
      int best = s->current_sync;
      for (int k = 0; k < s->nsyncs; ++k) if (total[k] > total[best]) best = k;
      if (getenv("LO_VIT_DEBUG")) {
        fprintf(stderr, "VIT out=%zu cur=%d best=%d :", nout_bytes, s->current_sync, best);
        for (int k = 0; k < s->nsyncs; ++k) fprintf(stderr, " %d", total[k]);
        fprintf(stderr, "\n");
      }
      s->current_sync = best;
    }
    if (++s->resync_phase >= s->resync_period) s->resync_phase = 0;
  }
  *consumed = pos;
  return nout_bytes;
}

/* ======================================================================= mpeg_sync, dvb.h:712-891 */
struct lo_mpeg_sync {
  int scan_syncs, want_syncs; unsigned long lock_timeout; int fastlock, resync_period;
  unsigned char polarity; int resync_phase, bitphase, synchronized, next_sync_count, phase8;
  unsigned long lock_timeleft, locktime; int report_state;
};
lo_mpeg_sync *lo_mpeg_sync_new(int fastlock) {
  lo_mpeg_sync *m = (lo_mpeg_sync *)calloc(1, sizeof(*m));
  m->scan_syncs = 8; m->want_syncs = 4; m->lock_timeout = 4; m->fastlock = fastlock; m->resync_period = 1;
  m->report_state = 1;
  return m;
}
void lo_mpeg_sync_free(lo_mpeg_sync *m) { free(m); }
void lo_mpeg_sync_set_resync_period(lo_mpeg_sync *m, int p) { m->resync_period = p; }
int lo_mpeg_sync_locked(const lo_mpeg_sync *m) { return m->synchronized; }

typedef struct { const uint8_t *in; size_t n_in, pos; uint8_t *out; size_t cap, nout; int *st; size_t st_cap, nst;
                 unsigned long *lt; size_t lt_cap, nlt; } ms_io;
/* dvb.h:798-840 */
static int ms_search_sync(lo_mpeg_sync *m, ms_io *io) {
  int chunk = SIZE_RSPACKET * m->scan_syncs;
  const uint8_t *pin = io->in + io->pos, *pend = pin + chunk;
  uint8_t *tmp = io->out + io->nout, *pout = tmp;
  unsigned short w = *pin++;
  for (; pin <= pend; ++pin, ++pout) { w = (w << 8) | *pin; *pout = w >> m->bitphase; }
  for (int i = 0; i < SIZE_RSPACKET; ++i) {
    int nsyncs_p = 0, nsyncs_n = 0, phase8_p = -1, phase8_n = -1;
    const uint8_t *p = &tmp[i];
    for (int j = 0; j < m->scan_syncs; ++j, p += SIZE_RSPACKET) {
      uint8_t b = *p;
      if (b == MPEG_SYNC) { ++nsyncs_p; phase8_n = (8 - j) & 7; }
      if (b == MPEG_SYNC_INV) { ++nsyncs_n; phase8_p = (8 - j) & 7; }
    }
    int nsyncs;
    if (nsyncs_p > nsyncs_n) { m->polarity = 0; nsyncs = nsyncs_p; m->phase8 = phase8_p; }
    else { m->polarity = 0xff; nsyncs = nsyncs_n; m->phase8 = phase8_n; }
    if (nsyncs >= m->want_syncs && m->phase8 >= 0) {
      if (!i) { i = SIZE_RSPACKET; m->phase8 = (m->phase8 + 1) & 7; }
      io->pos += i;
      m->synchronized = 1;
      m->lock_timeleft = m->lock_timeout;
      m->locktime = 0;
      if (io->st) io->st[io->nst++] = 1;
      return 1;
    }
  }
  return 0;
}
/* One run() call (dvb.h:743-754).  *call_next_sync = 1 when the reference would call deconv->next_sync(). */
size_t lo_mpeg_sync_run(lo_mpeg_sync *m, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed,
                        int *state_out, size_t state_cap, size_t *n_state,
                        unsigned long *locktime_out, size_t lt_cap, size_t *n_lt, int *call_next_sync) {
  ms_io io = {in, n_in, 0, out, cap, 0, state_out, state_cap, 0, locktime_out, lt_cap, 0};
  *call_next_sync = 0;
  if (m->report_state && state_out && state_cap >= 1) { state_out[io.nst++] = 0; m->report_state = 0; }
  const int chunk = SIZE_RSPACKET * m->scan_syncs;
#define ST_ROOM (!state_out || state_cap - io.nst >= 1)
  if (m->synchronized) {   /* run_decoding, dvb.h:842-875 */
    while (n_in - io.pos >= SIZE_RSPACKET + 1 && cap - io.nout >= SIZE_RSPACKET && ST_ROOM &&
           (!locktime_out || lt_cap - io.nlt >= 1)) {
      const uint8_t *pin = in + io.pos, *pend = pin + SIZE_RSPACKET;
      uint8_t *pout = out + io.nout;
      unsigned short w = *pin++;
      for (; pin <= pend; ++pin, ++pout) { w = (w << 8) | *pin; *pout = (w >> m->bitphase) ^ m->polarity; }
      io.pos += SIZE_RSPACKET;
      uint8_t syncbyte = out[io.nout];
      io.nout += SIZE_RSPACKET;
      ++m->locktime;
      if (locktime_out) locktime_out[io.nlt++] = m->locktime;
      uint8_t expected = m->phase8 ? MPEG_SYNC : MPEG_SYNC_INV;
      if (syncbyte == expected) m->lock_timeleft = m->lock_timeout;
      m->phase8 = (m->phase8 + 1) & 7;
      --m->lock_timeleft;
      if (!m->lock_timeleft) {
        m->synchronized = 0;
        m->next_sync_count = 0;
        if (state_out) state_out[io.nst++] = 0;
        break;
      }
    }
  } else if (m->fastlock) {   /* run_searching_fast, dvb.h:782-796 */
    int done = 0;
    while (!done && n_in - io.pos >= (size_t)chunk + 1 && cap - io.nout >= (size_t)chunk && ST_ROOM) {
      if (m->resync_phase == 0)
        for (m->bitphase = 0; m->bitphase <= 7; ++m->bitphase)
          if (ms_search_sync(m, &io)) { done = 1; break; }
      if (done) break;
      io.pos += SIZE_RSPACKET;
      if (++m->resync_phase >= m->resync_period) m->resync_phase = 0;
    }
  } else {   /* run_searching, dvb.h:756-780 */
    int next_sync = 0, locked_now = 0;
    while (n_in - io.pos >= (size_t)chunk + 1 && cap - io.nout >= (size_t)chunk && ST_ROOM) {
      if (ms_search_sync(m, &io)) { locked_now = 1; break; }
      io.pos += chunk;
      ++m->bitphase;
      if (m->bitphase == 8) { m->bitphase = 0; next_sync = 1; }
    }
    if (!locked_now && next_sync) {
      ++m->next_sync_count;
      if (m->next_sync_count >= 3) { m->next_sync_count = 0; *call_next_sync = 1; }
    }
  }
#undef ST_ROOM
  *consumed = io.pos;
  if (n_state) *n_state = io.nst;
  if (n_lt) *n_lt = io.nlt;
  return io.nout;
}

/* ======================================================================= deinterleaver, dvb.h:926-948 */
size_t lo_deinterleaver(const uint8_t *in, size_t n_in, uint8_t *out, size_t cap_packets, size_t *consumed) {
  size_t pos = 0, np = 0;
  while (n_in - pos >= 17 * 11 * 12 + SIZE_RSPACKET && n_in >= pos && np < cap_packets) {
    const uint8_t *pin = in + pos + 17 * 11 * 12, *pend = pin + SIZE_RSPACKET;
    uint8_t *pout = out + np * SIZE_RSPACKET;
    for (int delay = 17 * 11; pin < pend; ++pin, ++pout, delay = (delay - 17 + 17 * 12) % (17 * 12)) *pout = pin[-delay * 12];
    pos += SIZE_RSPACKET;
    ++np;
  }
  *consumed = pos;
  return np;
}

/* ======================================================================= RS(204,188), rs.h:47-272 */
static uint8_t gf_exp[512], gf_log[256], rs_G[17];
static int gf_ready = 0;
static void gf_init(void) {
  if (gf_ready) return;
  unsigned a = 1;
  memset(gf_log, 0, sizeof(gf_log));   /* lut_log[0] is never written by the reference: fresh-heap zero */
  for (unsigned i = 0; i < 256; ++i) {
    gf_exp[i] = a; gf_exp[255 + i] = a; gf_log[a] = i;
    a <<= 1; if (a & 256) a ^= 0x11d;
  }
  gf_ready = 1;
}
static inline uint8_t gmul(uint8_t x, uint8_t y) { return (!x || !y) ? 0 : gf_exp[gf_log[x] + gf_log[y]]; }
static inline uint8_t gdiv(uint8_t x, uint8_t y) { return !x ? 0 : gf_exp[gf_log[x] + 255 - gf_log[y]]; }
static inline uint8_t ginv(uint8_t x) { return gf_exp[255 - gf_log[x]]; }
static void rs_init(void) {
  gf_init();
  for (int i = 0; i <= 16; ++i) rs_G[i] = (i == 16) ? 1 : 0;
  for (int d = 0; d < 16; ++d)
    for (int i = 0; i <= 16; ++i) rs_G[i] = ((i == 16) ? 0 : rs_G[i + 1]) ^ gmul(gf_exp[d], rs_G[i]);
}
void lo_rs_tables(uint8_t *exp512, uint8_t *log256, uint8_t *G17) {
  rs_init();
  for (int i = 0; i < 512; ++i) exp512[i] = gf_exp[i < 510 ? i : 0];
  memcpy(log256, gf_log, 256); memcpy(G17, rs_G, 17);
}
static uint8_t eval_poly_rev(const uint8_t *p, int n, uint8_t x) { uint8_t a = 0; for (int i = 0; i < n; ++i) a = gmul(a, x) ^ p[i]; return a; }
static uint8_t eval_poly(const uint8_t *p, int deg, uint8_t x) { uint8_t a = 0; for (; deg >= 0; --deg) a = gmul(a, x) ^ p[deg]; return a; }
static int rs_syndromes(const uint8_t *poly, uint8_t *synd) {   /* rs.h:116-123 */
  int corrupted = 0;
  for (int i = 0; i < 16; ++i) { synd[i] = eval_poly_rev(poly, 204, gf_exp[i]); if (synd[i]) corrupted = 1; }
  return corrupted;
}
void lo_rs_encode(uint8_t *msg) {   /* rs.h:141-167 */
  rs_init();
  uint8_t p[204];
  memcpy(p, msg, 188); memset(p + 188, 0, 16);
  for (int d = 0; d < 188; ++d) {
    if (!p[d]) continue;
    uint8_t k = gdiv(p[d], rs_G[0]);
    for (int i = 0; i <= 16; ++i) p[d + i] ^= gmul(k, rs_G[i]);
  }
  memcpy(msg + 188, p + 188, 16);
}
/* rs_engine::correct, rs.h:173-270 */
static int rs_correct(uint8_t synd[16], uint8_t pout[188], uint8_t pin[204], int *bits_corrected) {
  uint8_t C[16] = {1}, B[16] = {1};
  int L = 0, m = 1; uint8_t b = 1;
  for (int n = 0; n < 16; ++n) {
    uint8_t d = synd[n];
    for (int i = 1; i <= L; ++i) d ^= gmul(C[i], synd[n - i]);
    if (!d) ++m;
    else if (2 * L <= n) {
      uint8_t T[16]; memcpy(T, C, 16);
      for (int i = 0; i < 16 - m; ++i) C[m + i] ^= gmul(d, gmul(ginv(b), B[i]));
      L = n + 1 - L; memcpy(B, T, 16); b = d; m = 1;
    } else {
      for (int i = 0; i < 16 - m; ++i) C[m + i] ^= gmul(d, gmul(ginv(b), B[i]));
      ++m;
    }
  }
  uint8_t omega[16]; memset(omega, 0, 16);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (i + j < 16) omega[i + j] ^= gmul(synd[i], C[j]);
  uint8_t Cprime[15];
  for (int i = 0; i < 15; ++i) Cprime[i] = (i & 1) ? 0 : C[i + 1];
  int roots_found = 0;
  for (int i = 0; i < 255; ++i) {
    uint8_t r = gf_exp[i];
    uint8_t v = eval_poly(C, L, r);
    if (!v) {
      uint8_t xk = ginv(r);
      int loc = (255 - i) % 255;
      if (loc < 204) {
        uint8_t num = gmul(xk, eval_poly(omega, L, r));
        uint8_t den = eval_poly(Cprime, 14, r);
        uint8_t e = gdiv(num, den);
        if (bits_corrected) *bits_corrected += __builtin_popcount(e);
        if (loc >= 16) pout[203 - loc] ^= e;
        pin[203 - loc] ^= e;
      }
      if (++roots_found == L) break;
    }
  }
  return rs_syndromes(pin, synd);
}
/* rs_decoder::run, dvb.h:998-1053 over npackets; the input is corrected in place like the reference */
size_t lo_rs_decoder(uint8_t *in, size_t npackets, uint8_t *out, long *bits, long *errs) {
  rs_init();
  int nerrs = 0; long nbits = 0;
  for (size_t p = 0; p < npackets; ++p) {
    uint8_t *pin = in + p * SIZE_RSPACKET, *pout = out + p * SIZE_TSPACKET;
    nbits += SIZE_RSPACKET * 8;
    memcpy(pout, pin, SIZE_TSPACKET);
    uint8_t synd[16];
    int corrupted = rs_syndromes(pin, synd);
    if (corrupted) corrupted = rs_correct(synd, pout, pin, &nerrs);
    if (corrupted) pout[0] ^= MPEG_SYNC_CORRUPTED;
  }
  if (bits) *bits = nbits;
  if (errs) *errs = nerrs;
  return npackets;
}

/* ======================================================================= derandomizer, dvb.h:1107-1163 */
void lo_derandomizer_pattern(uint8_t *pattern /*[1504]*/) {
  pattern[0] = 0xff;
  unsigned short st = 000251;
  for (int i = 1; i < 188 * 8; ++i) {
    uint8_t o = 0;
    for (int n = 8; n--;) { int bit = ((st >> 13) ^ (st >> 14)) & 1; o = (o << 1) | bit; st = (st << 1) | bit; }
    pattern[i] = (i % 188) ? o : 0;
  }
}
struct lo_derandomizer { uint8_t pattern[1504]; int pos; };
lo_derandomizer *lo_derandomizer_new(void) { lo_derandomizer *d = calloc(1, sizeof(*d)); lo_derandomizer_pattern(d->pattern); return d; }
void lo_derandomizer_free(lo_derandomizer *d) { free(d); }
size_t lo_derandomizer_run(lo_derandomizer *d, const uint8_t *in, size_t npackets, uint8_t *out) {
  size_t nout = 0;
  for (size_t p = 0; p < npackets; ++p) {
    const uint8_t *pin = in + p * SIZE_TSPACKET;
    uint8_t *pout = out + nout * SIZE_TSPACKET;
    if (pin[0] == MPEG_SYNC_INV || pin[0] == (MPEG_SYNC_INV ^ MPEG_SYNC_CORRUPTED)) d->pos = 0;
    for (int i = 0; i < SIZE_TSPACKET; ++i) pout[i] = pin[i] ^ d->pattern[d->pos + i];
    d->pos += SIZE_TSPACKET;
    if (d->pos == 1504) d->pos = 0;
    if (pout[0] == MPEG_SYNC) ++nout;   /* otherwise TEI is set in a slot that is never committed */
  }
  return nout;
}

/* ======================================================================= whole tail, leandvb.cc:519-596
 * Scheduler-faithful emulation is not needed for a feed-forward check: run block after block over the
 * whole stream, re-invoking mpeg_sync/deconvol like successive run() calls until no progress. */
size_t lo_fec_chain(int cstln, int rate, int viterbi, int fastlock, const lo_softsymbol *sym, size_t n,
                    uint8_t *ts_out, size_t cap_packets, long *bits, long *errs) {
  uint8_t *bytes = malloc(n + 4096), *mpeg = malloc(n + 4096);
  size_t nbytes = 0, nmpeg = 0;
  lo_mpeg_sync *ms = lo_mpeg_sync_new(fastlock);
  if (viterbi) {
    lo_viterbi *v = lo_viterbi_new(cstln, rate);
    if (fastlock) v->resync_period = 1;
    size_t c;
    nbytes = lo_viterbi_run(v, sym, n, bytes, n + 4096, &c);
    lo_viterbi_free(v);
    size_t pos = 0;
    for (;;) {
      size_t cons; int cns;
      size_t got = lo_mpeg_sync_run(ms, bytes + pos, nbytes - pos, mpeg + nmpeg, n + 4096 - nmpeg, &cons, NULL, 0, NULL, NULL, 0, NULL, &cns);
      if (!got && !cons) break;
      pos += cons; nmpeg += got;
    }
  } else {
    /* deconvol_sync and mpeg_sync interact (next_sync): emulate the scheduler's alternation with the
     * reference's default pipe sizes (bytes 8192, symbols 4096, leandvb.cc:185-202, buf_factor 4). */
    lo_deconv *d = lo_deconv_new(rate, fastlock);
    size_t spos = 0, bpos = 0;   /* consumed symbols / bytes */
    for (;;) {
      size_t progress = 0;
      size_t avail = n - spos; if (avail > 4096) avail = 4096;
      size_t room = 8192 - (nbytes - bpos); if (room > 8192) room = 0;
      size_t c;
      size_t got = lo_deconv_run(d, sym + spos, avail, bytes + nbytes, room, &c);
      spos += c; nbytes += got; progress += c + got;
      size_t cons; int cns;
      size_t bavail = nbytes - bpos;
      got = lo_mpeg_sync_run(ms, bytes + bpos, bavail, mpeg + nmpeg, n + 4096 - nmpeg, &cons, NULL, 0, NULL, NULL, 0, NULL, &cns);
      if (cns) lo_deconv_next_sync(d);
      bpos += cons; nmpeg += got; progress += cons + got + (size_t)cns;
      if (!progress) break;
    }
    lo_deconv_free(d);
  }
  lo_mpeg_sync_free(ms);
  size_t npk_max = nmpeg / SIZE_RSPACKET + 1;
  uint8_t *rsp = malloc(npk_max * SIZE_RSPACKET), *rts = malloc(npk_max * SIZE_TSPACKET);
  size_t c;
  size_t npk = lo_deinterleaver(mpeg, nmpeg, rsp, npk_max, &c);
  lo_rs_decoder(rsp, npk, rts, bits, errs);
  lo_derandomizer *dr = lo_derandomizer_new();
  uint8_t *tmp = malloc((npk + 1) * SIZE_TSPACKET);
  size_t nts = lo_derandomizer_run(dr, rts, npk, tmp);
  if (nts > cap_packets) nts = cap_packets;
  memcpy(ts_out, tmp, nts * SIZE_TSPACKET);
  lo_derandomizer_free(dr);
  free(bytes); free(mpeg); free(rsp); free(rts); free(tmp);
  return nts;
}
