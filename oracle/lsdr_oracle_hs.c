/* oracle/lsdr_oracle_hs.c — CPU ORACLE (test infrastructure, see lsdr_oracle.h) for the `--hs` path of
 * leandvb (leandvb.cc:727-969): fast_qpsk_receiver<u8> (sdr.h:946-1189) and dvb_deconvol_sync<u8>
 * (dvb.h:612-707) on deconvol_poly2<…,0x3ba,0x38f70> (convolutional.h:80-192).  Integer state except the
 * symbol clock `mu` (float).  Every conversion is spelled out the way the reference's expressions convert. */
#include "lsdr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- fast_qpsk_receiver */
struct lo_fastqpsk {
  uint16_t polar_a[65536];     /* lut_polar[re][im].a, index re*256+im (sdr.h:1155-1160) */
  uint8_t polar_r[65536];      /*                 .r */
  uint8_t rect[256][256][2];   /* lut_rect[a][r] = {re, im} (sdr.h:1166-1170) */
  uint8_t sincos[65536][2];    /* lut_sincos[a] (sdr.h:1161-1165) */
  unsigned long meas_decimation;
  float omega, min_omega, max_omega;
  long freqw, min_freqw, max_freqw;
  float pll_adjustment;
  int allow_drift;
  float mu;
  uint16_t phase;
  unsigned long meas_count;
  uint8_t hist_p[3][2], hist_c[3][2];
};

static void fq_update_freq_limits(lo_fastqpsk *r) {          /* sdr.h:987-992 */
  r->min_freqw = (long)(r->freqw - 65536 / r->max_omega / 8);
  r->max_freqw = (long)(r->freqw + 65536 / r->max_omega / 8);
}
void lo_fastqpsk_set_omega(lo_fastqpsk *r, float omega) {     /* sdr.h:975-980, tol = 10e-6 */
  const float tol = 10e-6;
  r->omega = omega;
  r->min_omega = omega * (1 - tol);
  r->max_omega = omega * (1 + tol);
  fq_update_freq_limits(r);
}
void lo_fastqpsk_set_freq(lo_fastqpsk *r, float freq) {       /* sdr.h:982-985 */
  r->freqw = (long)(freq * 65536);
  fq_update_freq_limits(r);
}

lo_fastqpsk *lo_fastqpsk_new(float omega, float freq, float pll_adjustment, int allow_drift, unsigned long meas_decimation) {
  lo_fastqpsk *r = (lo_fastqpsk *)calloc(1, sizeof(*r));
  r->meas_decimation = meas_decimation ? meas_decimation : 1048576;
  r->pll_adjustment = pll_adjustment;
  r->allow_drift = allow_drift;
  r->mu = 0; r->phase = 0; r->meas_count = 0;
  lo_fastqpsk_set_omega(r, 1);
  lo_fastqpsk_set_freq(r, 0);
  /* init_lookup_tables, sdr.h:1154-1171 */
  for (int i = 0; i < 256; ++i)
    for (int q = 0; q < 256; ++q) {
      /* (s_angle)(atan2f(q-128,i-128)*65536/(2*M_PI)): float·int → float, / double → double, → int16.
       * +π gives 32768.0…: out of int16 range; the reference build wraps (via int32) to -32768. */
      double v = atan2f((float)(q - 128), (float)(i - 128)) * 65536 / (2 * M_PI);
      r->polar_a[i * 256 + q] = (uint16_t)(int16_t)(int32_t)v;
      r->polar_r[i * 256 + q] = (uint8_t)(int)hypotf((float)(i - 128), (float)(q - 128));
    }
  for (unsigned long a = 0; a < 65536; ++a) {
    float f = 2 * M_PI * a / 65536;
    r->sincos[a][0] = (uint8_t)(128 + 75.0f * cosf(f));
    r->sincos[a][1] = (uint8_t)(128 + 75.0f * sinf(f));
  }
  for (int a = 0; a < 256; ++a)
    for (int rr = 0; rr < 256; ++rr) {
      r->rect[a][rr][0] = (uint8_t)(int)(128 + rr * cos(2 * M_PI * a / 256));
      r->rect[a][rr][1] = (uint8_t)(int)(128 + rr * sin(2 * M_PI * a / 256));
    }
  if (omega) lo_fastqpsk_set_omega(r, omega);
  if (freq) lo_fastqpsk_set_freq(r, freq);
  return r;
}
void lo_fastqpsk_free(lo_fastqpsk *r) { free(r); }
void lo_fastqpsk_tables(const lo_fastqpsk *r, uint16_t *polar_a, uint8_t *polar_r, uint8_t *rect, uint8_t *sincos) {
  memcpy(polar_a, r->polar_a, sizeof(r->polar_a));
  memcpy(polar_r, r->polar_r, sizeof(r->polar_r));
  memcpy(rect, r->rect, sizeof(r->rect));
  memcpy(sincos, r->sincos, sizeof(r->sincos));
}
void lo_fastqpsk_get_state(const lo_fastqpsk *r, float *mu, unsigned *phase, long *freqw, long *min_freqw, long *max_freqw) {
  *mu = r->mu; *phase = r->phase; *freqw = r->freqw; *min_freqw = r->min_freqw; *max_freqw = r->max_freqw;
}

/* run(), sdr.h:997-1140.  in: cu8 samples; out: hard symbols; freq_out / cstln_out may be NULL. */
size_t lo_fastqpsk_run(lo_fastqpsk *r, const lo_cu8 *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed,
                       float *freq_out, size_t freq_cap, size_t *n_freq, lo_cu8 *cstln_out, size_t cstln_cap, size_t *n_cstln) {
  const long freq_alpha = (long)(0.04 * 65536);
  const long freq_beta = (long)(0.0012 * 256 * 65536 / r->omega * r->pll_adjustment);
  const float gain_mu = 0.02 / (75.0f * 75.0f) * 2;
  const size_t max_meas = 128 / r->meas_decimation + 1;
  size_t pos = 0, nout = 0, nf = 0, nc = 0;
  static const unsigned char quadrant_to_symbol[4] = {0, 2, 3, 1};
  while (n_in - pos >= 129 && cap - nout >= 128 && (!freq_out || freq_cap - nf >= max_meas) &&
         (!cstln_out || cstln_cap - nc >= max_meas)) {
    const lo_cu8 *pin = in + pos, *pend = pin + 128;
    lo_cu8 s = {0, 0};
    uint16_t symbol_arg = 0;
    while (pin < pend) {
      if (r->mu < 1) {
        const unsigned i0 = pin[0].re * 256u + pin[0].im, i1 = pin[1].re * 256u + pin[1].im;
        const uint16_t a0 = (uint16_t)((uint16_t)(r->polar_a[i0] - r->phase) >> 8);
        const uint8_t *p0r = r->rect[a0][r->polar_r[i0] >> 1];
        const uint16_t a1 = (uint16_t)((uint16_t)((long)r->polar_a[i1] - ((long)r->phase + r->freqw)) >> 8);
        const uint8_t *p1r = r->rect[a1][r->polar_r[i1] >> 1];
        s.re = (uint8_t)(int)(p0r[0] + (p1r[0] - p0r[0]) * r->mu);
        s.im = (uint8_t)(int)(p0r[1] + (p1r[1] - p0r[1]) * r->mu);
        symbol_arg = r->polar_a[s.re * 256u + s.im];
        out[nout++] = quadrant_to_symbol[symbol_arg >> 14];
        /* PLL, sdr.h:1072-1074 */
        const int16_t phase_error = (int16_t)((int16_t)(symbol_arg & 16383) - 8192);
        r->phase = (uint16_t)(r->phase + ((phase_error * freq_alpha + 32768) >> 16));
        r->freqw += (phase_error * freq_beta + 32768 * 256) >> 24;
        /* Modified Mueller & Müller on cu8 history, sdr.h:1081-1105 */
        memcpy(r->hist_p[2], r->hist_p[1], 2); memcpy(r->hist_c[2], r->hist_c[1], 2);
        memcpy(r->hist_p[1], r->hist_p[0], 2); memcpy(r->hist_c[1], r->hist_c[0], 2);
        r->hist_p[0][0] = s.re; r->hist_p[0][1] = s.im;
        const uint16_t ca = (uint16_t)((symbol_arg & 49152) + 8192);
        r->hist_c[0][0] = r->sincos[ca][0]; r->hist_c[0][1] = r->sincos[ca][1];
        const int muerr =
            ((signed char)(r->hist_p[0][0] - r->hist_p[2][0]) * ((int)r->hist_c[1][0] - 128) +
             (signed char)(r->hist_p[0][1] - r->hist_p[2][1]) * ((int)r->hist_c[1][1] - 128)) -
            ((signed char)(r->hist_c[0][0] - r->hist_c[2][0]) * ((int)r->hist_p[1][0] - 128) +
             (signed char)(r->hist_c[0][1] - r->hist_c[2][1]) * ((int)r->hist_p[1][1] - 128));
        float mucorr = muerr * gain_mu;
        const float max_mucorr = 0.1;
        if (mucorr < -max_mucorr) mucorr = -max_mucorr;
        if (mucorr > max_mucorr) mucorr = max_mucorr;
        r->mu += mucorr;
        r->mu += r->omega;
      }
      ++pin;
      --r->mu;
      r->phase = (uint16_t)(r->phase + r->freqw);
    }
    pos += 128;
    if (symbol_arg && cstln_out) cstln_out[nc++] = s;
    if (!r->allow_drift)
      if (r->freqw < r->min_freqw || r->freqw > r->max_freqw) r->freqw = (r->max_freqw + r->min_freqw) / 2;
    r->meas_count += 128;
    while (r->meas_count >= r->meas_decimation) {
      r->meas_count -= r->meas_decimation;
      if (freq_out) freq_out[nf++] = (float)r->freqw / 65536;
    }
  }
  *consumed = pos;
  if (n_freq) *n_freq = nf;
  if (n_cstln) *n_cstln = nc;
  return nout;
}

/* ---------------------------------------------------------------- dvb_deconvol_sync<u8> */
struct lo_hsdeconv {
  int resync_period, resync_phase, locked;
  uint32_t inI[4], inQ[4];
  uint8_t lut[4][4];
};
lo_hsdeconv *lo_hsdeconv_new(int resync_period) {
  static const uint8_t luts[4][4] = {{0, 1, 2, 3}, {2, 0, 3, 1}, {1, 0, 3, 2}, {0, 2, 1, 3}};   /* dvb.h:676-699 */
  lo_hsdeconv *d = (lo_hsdeconv *)calloc(1, sizeof(*d));
  d->resync_period = resync_period; d->resync_phase = 0; d->locked = 0;
  memcpy(d->lut, luts, sizeof(luts));
  return d;
}
void lo_hsdeconv_free(lo_hsdeconv *d) { free(d); }
int lo_hsdeconv_locked(const lo_hsdeconv *d) { return d->locked; }

/* deconvol_poly2<u8,uint32_t,uint64_t,0x3ba,0x38f70>::run over nb = 64 bytes, convolutional.h:94-189 */
static int hs_poly2_run(uint32_t *inI, uint32_t *inQ, const uint8_t *pin, const uint8_t *remap, uint8_t *pout) {
  const uint64_t PD = 0x3ba, PE = 0x38f70;
  int nb = 64 / 4;
  unsigned long nerrors = 0;
  const int halfway = nb / 2;
  uint32_t histI = *inI, histQ = *inQ;
  for (; nb--;) {
    uint32_t wd = 0, we = 0;
    for (int bit = 31; bit >= 0; --bit, ++pin) {
      const uint8_t iq = remap[*pin];
      histI = (histI << 1) | (iq >> 1);
      histQ = (histQ << 1) | (iq & 1);
      if (PD & ((uint64_t)2 << (2 * bit))) wd ^= histI;
      if (PD & ((uint64_t)1 << (2 * bit))) wd ^= histQ;
      if (PE & ((uint64_t)2 << (2 * bit))) we ^= histI;
      if (PE & ((uint64_t)1 << (2 * bit))) we ^= histQ;
    }
    *pout++ = wd >> 24; *pout++ = wd >> 16; *pout++ = wd >> 8; *pout++ = wd;
    if (nb < halfway) nerrors += __builtin_popcount(we);
  }
  *inI = histI; *inQ = histQ;
  return (int)nerrors;
}

/* run(), dvb.h:634-660 */
size_t lo_hsdeconv_run(lo_hsdeconv *d, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed) {
  size_t pos = 0, nout = 0;
  uint8_t dummy[64];
  while (n_in - pos >= 64 * 8 && cap - nout >= 64) {
    int errors_best = 1 << 30, best = -1;
    for (int s = 0; s < 4; ++s) {
      if (d->resync_phase != 0 && s != d->locked) continue;
      uint8_t *pout = s == d->locked ? out + nout : dummy;
      const int nerrors = hs_poly2_run(&d->inI[s], &d->inQ[s], in + pos, d->lut[s], pout);
      if (nerrors < errors_best) { errors_best = nerrors; best = s; }
    }
    pos += 64 * 8;
    nout += 64;
    if (best != d->locked) d->locked = best;
    if (++d->resync_phase >= d->resync_period) d->resync_phase = 0;
  }
  *consumed = pos;
  return nout;
}
