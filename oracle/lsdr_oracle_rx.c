/* oracle/lsdr_oracle_rx.c — CPU ORACLE (test infrastructure).
 * cstln_receiver<f32> with its three samplers (sdr.h:589-938).
 * Strictly sequential, one IEEE float op per source op (-ffp-contract=off). */
#include "lsdr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CHUNK 128 /* sdr.h:706 */
static const float cstln_amp = 75; /* sdr.h:297 */

struct lo_rx {
  lo_rx_params p;
  lo_cf32 *trig;          /* trig16 (each reference sampler owns one, sdr.h:607,627,684) */
  lo_cstln_lut *cstln;
  /* fir_sampler state, sdr.h:635-689 */
  lo_cf32 *shifted;
  int update_freq_phase;
  /* linear_sampler state, sdr.h:625-628 */
  float samp_freqw;
  /* receiver state, sdr.h:923-935 */
  float omega, min_omega, max_omega;
  float freqw, min_freqw, max_freqw;
  float est_insp, agc_gain, mu, phase, est_sp, est_ep, freq_tap;
  unsigned long meas_count;
  struct { lo_cf32 p, c; } hist[3];
};

/* sdr.h:755-770 */
static void update_freq_limits(lo_rx *r) {
  int n = 4;
  if (r->cstln) {
    switch (r->cstln->nsymbols) {
      case 2: n = 2; break;
      case 4: n = 4; break;
      case 8: n = 8; break;
      case 16: n = 12; break;
      case 32: n = 16; break;
      default: n = 4; break;
    }
  }
  r->min_freqw = r->freqw - 65536 / r->max_omega / n / 2;
  r->max_freqw = r->freqw + 65536 / r->max_omega / n / 2;
}

/* sdr.h:738-743; tol = 10e-6 (double literal converted to float parameter) */
static void set_omega(lo_rx *r, float omega) {
  float tol = 10e-6;
  r->omega = omega;
  r->min_omega = omega * (1 - tol);
  r->max_omega = omega * (1 + tol);
  update_freq_limits(r);
}

/* sdr.h:745-749 */
static void set_freq(lo_rx *r, float freq) {
  r->freqw = freq * 65536;
  update_freq_limits(r);
  r->freq_tap = r->freqw / 65536;
}

/* Construction order follows leandvb.cc:463-502: ctor (cstln==NULL:
 * set_omega(1), set_freq(0)) -> cstln assigned -> set_omega(Fs/Fm) ->
 * optional set_freq(). */
lo_rx *lo_rx_new(const lo_rx_params *p) {
  lo_rx *r = (lo_rx *)calloc(1, sizeof(*r));
  r->p = *p;
  r->trig = (lo_cf32 *)malloc(sizeof(lo_cf32) * 65536);
  lo_trig16(r->trig);
  r->est_insp = cstln_amp * cstln_amp; r->agc_gain = 1;
  r->mu = 0; r->phase = 0; r->est_sp = 0; r->est_ep = 0; r->meas_count = 0;
  r->cstln = NULL;
  set_omega(r, 1);
  set_freq(r, 0);
  r->cstln = (lo_cstln_lut *)malloc(sizeof(lo_cstln_lut));
  lo_make_dvbs2_constellation(r->cstln, p->cstln, p->fec);
  set_omega(r, p->omega);
  if (p->freq) set_freq(r, p->freq);
  if (p->sampler == LO_SAMP_FIR) {
    r->shifted = (lo_cf32 *)calloc(p->ncoeffs, sizeof(lo_cf32));
    r->update_freq_phase = 0;
  }
  return r;
}

void lo_rx_free(lo_rx *r) { free(r->trig); free(r->cstln); free(r->shifted); free(r); }

int lo_rx_readahead(const lo_rx *r) {
  switch (r->p.sampler) {
    case LO_SAMP_NEAREST: return 0;
    case LO_SAMP_LINEAR: return 1;
    default: return r->p.ncoeffs - 1;
  }
}

static inline lo_cf32 cmul(lo_cf32 a, lo_cf32 b) { /* math.h:40-43 */
  lo_cf32 r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
  return r;
}
static inline lo_cf32 expi(const lo_rx *r, float a) { return r->trig[lo_trig16_index(a)]; }

/* fir_sampler::do_update_freq, sdr.h:676-680: (i-ncoeffs/2) is int here. */
static void fir_do_update_freq(lo_rx *r, float freqw) {
  float f = freqw / r->p.subsampling;
  int N = r->p.ncoeffs;
  for (int i = 0; i < N; ++i) {
    lo_cf32 e = expi(r, -f * (i - N / 2));
    r->shifted[i].re = e.re * r->p.coeffs[i]; /* complex*T, math.h:45-48 */
    r->shifted[i].im = e.im * r->p.coeffs[i];
  }
}

/* sampler_interface::update_freq: sdr.h:625 (linear), sdr.h:667-674 (fir) */
static void sampler_update_freq(lo_rx *r, float freqw) {
  if (r->p.sampler == LO_SAMP_LINEAR) r->samp_freqw = freqw;
  else if (r->p.sampler == LO_SAMP_FIR) {
    r->update_freq_phase -= 128;
    if (r->update_freq_phase <= 0) {
      r->update_freq_phase = r->p.ncoeffs * 16;
      fir_do_update_freq(r, freqw);
    }
  }
}

/* sdr.h:602-604, 614-623, 646-665 */
static lo_cf32 sampler_interp(const lo_rx *r, const lo_cf32 *pin, float mu, float phase) {
  if (r->p.sampler == LO_SAMP_NEAREST) return cmul(pin[0], expi(r, -phase));
  if (r->p.sampler == LO_SAMP_LINEAR) {
    lo_cf32 s0 = cmul(pin[0], expi(r, -phase));
    lo_cf32 s1 = cmul(pin[1], expi(r, -(phase + r->samp_freqw)));
    float k0 = 1 - mu;
    lo_cf32 o = {s0.re * k0 + s1.re * mu, s0.im * k0 + s1.im * mu};
    return o;
  }
  lo_cf32 acc = {0, 0};
  int S = r->p.subsampling, N = r->p.ncoeffs;
  for (int pc = (int)((1 - mu) * S); pc < N; pc += S, ++pin) {
    lo_cf32 t = cmul(r->shifted[pc], *pin);
    acc.re += t.re;
    acc.im += t.im;
  }
  return cmul(expi(r, -phase), acc);
}

/* sdr.h:772-916 */
size_t lo_rx_run(lo_rx *r, const lo_cf32 *in, size_t n_in,
                 lo_softsymbol *out, size_t cap, size_t *consumed,
                 float *freq_out, float *ss_out, float *mer_out, size_t meas_cap, size_t *n_meas,
                 lo_cf32 *cstln_out, size_t cstln_cap, size_t *n_cstln) {
  const lo_cstln_lut *L = r->cstln;
  float freq_alpha = 0.04;
  float freq_beta = 0.0012 / r->omega * r->p.pll_adjustment; /* double expr -> float */
  float gain_mu = 0.02 / (cstln_amp * cstln_amp) * 2;
  size_t pos = 0, nout = 0, nm = 0, nc = 0;
  size_t max_meas = CHUNK / r->p.meas_decimation + 1;
  int ra = lo_rx_readahead(r);

  while (n_in - pos >= (size_t)(CHUNK + ra) && n_in >= pos && cap - nout >= CHUNK &&
         (!freq_out || meas_cap - nm >= max_meas) &&
         (!cstln_out || cstln_cap - nc >= max_meas)) {
    sampler_update_freq(r, r->freqw);
    const lo_cf32 *pin = in + pos, *pend = pin + CHUNK;
    lo_cf32 sg = {0, 0}, s = {0, 0};
    const int8_t *cstln_point = NULL;

    while (pin < pend) {
      if (r->mu < 1) {
        sg = sampler_interp(r, pin, r->mu, r->phase);
        s.re = sg.re * r->agc_gain;
        s.im = sg.im * r->agc_gain;
        unsigned idx = lo_cstln_lookup_index(s.re, s.im);
        out[nout].cost = L->cost[idx];
        out[nout].symbol = L->symbol[idx];
        out[nout].pad = 0;
        ++nout;
        int16_t pe = L->phase_error[idx];
        r->phase += pe * freq_alpha;
        r->freqw += pe * freq_beta;
        r->hist[2] = r->hist[1];
        r->hist[1] = r->hist[0];
        r->hist[0].p.re = s.re;
        r->hist[0].p.im = s.im;
        cstln_point = L->symbols[L->symbol[idx]];
        r->hist[0].c.re = cstln_point[0];
        r->hist[0].c.im = cstln_point[1];
        float muerr =
            ((r->hist[0].p.re - r->hist[2].p.re) * r->hist[1].c.re +
             (r->hist[0].p.im - r->hist[2].p.im) * r->hist[1].c.im) -
            ((r->hist[0].c.re - r->hist[2].c.re) * r->hist[1].p.re +
             (r->hist[0].c.im - r->hist[2].c.im) * r->hist[1].p.im);
        float mucorr = muerr * gain_mu;
        const float max_mucorr = 0.1;
        if (mucorr < -max_mucorr) mucorr = -max_mucorr;
        if (mucorr > max_mucorr) mucorr = max_mucorr;
        r->mu += mucorr;
        r->mu += r->omega;
      }
      ++pin;
      --r->mu;
      r->phase += r->freqw;
    }
    pos += CHUNK;
    r->phase = fmodf(r->phase, 65536);

    if (cstln_point) {
      if (cstln_out) cstln_out[nc++] = s;
      float insp = sg.re * sg.re + sg.im * sg.im;
      r->est_insp = insp * r->p.kest + r->est_insp * (1 - r->p.kest);
      if (r->est_insp) r->agc_gain = cstln_amp / sqrtf(r->est_insp);
      float evre = s.re - cstln_point[0], evim = s.im - cstln_point[1];
      float sig_power, ev_power;
      if (L->nsymbols == 2) {
        /* (re+im) int, *0.707 double -> float (sdr.h:877-878) */
        float sig_real = (cstln_point[0] + cstln_point[1]) * 0.707;
        float ev_real = (evre + evim) * 0.707;
        sig_power = sig_real * sig_real;
        ev_power = ev_real * ev_real;
      } else {
        sig_power = (int)cstln_point[0] * cstln_point[0] + (int)cstln_point[1] * cstln_point[1];
        ev_power = evre * evre + evim * evim;
      }
      r->est_sp = sig_power * r->p.kest + r->est_sp * (1 - r->p.kest);
      r->est_ep = ev_power * r->p.kest + r->est_ep * (1 - r->p.kest);
    }
    if (!r->p.allow_drift) {
      if (r->freqw < r->min_freqw || r->freqw > r->max_freqw)
        r->freqw = (r->max_freqw + r->min_freqw) / 2;
    }
    r->freq_tap = r->freqw / 65536;
    r->meas_count += CHUNK;
    while (r->meas_count >= r->p.meas_decimation) {
      r->meas_count -= r->p.meas_decimation;
      if (freq_out) freq_out[nm] = r->freq_tap;
      if (ss_out) ss_out[nm] = sqrtf(r->est_insp);
      if (mer_out) mer_out[nm] = r->est_ep ? 10 * logf(r->est_sp / r->est_ep) / logf(10) : 0;
      ++nm;
    }
  }
  if (consumed) *consumed = pos;
  if (n_meas) *n_meas = nm;
  if (n_cstln) *n_cstln = nc;
  return nout;
}

void lo_rx_get_state(const lo_rx *r, lo_rx_state *st) {
  st->mu = r->mu; st->phase = r->phase; st->freqw = r->freqw; st->agc_gain = r->agc_gain;
  st->est_insp = r->est_insp; st->est_sp = r->est_sp; st->est_ep = r->est_ep;
  st->freq_tap = r->freq_tap; st->min_freqw = r->min_freqw; st->max_freqw = r->max_freqw;
  st->meas_count = r->meas_count;
  memcpy(st->hist, r->hist, sizeof(st->hist));
}
void lo_rx_set_state(lo_rx *r, const lo_rx_state *st) {
  r->mu = st->mu; r->phase = st->phase; r->freqw = st->freqw; r->agc_gain = st->agc_gain;
  r->est_insp = st->est_insp; r->est_sp = st->est_sp; r->est_ep = st->est_ep;
  r->freq_tap = st->freq_tap; r->min_freqw = st->min_freqw; r->max_freqw = st->max_freqw;
  r->meas_count = st->meas_count;
  memcpy(r->hist, st->hist, sizeof(st->hist));
}
