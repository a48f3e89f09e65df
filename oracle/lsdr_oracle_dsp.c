/* oracle/lsdr_oracle_dsp.c — CPU ORACLE (test infrastructure).
 * Streaming DSP blocks: cconverter, scaler, decimator, fir_filter,
 * fir_resampler, cfft_engine, auto_notch, cnr_fft.
 * Build with -ffp-contract=off: every float op below is one IEEE operation in
 * source order, exactly as the reference's SSE2 code evaluates it. */
#include "lsdr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* dsp.h:40-50: out = Zout + (in - Zin)*Gn/Gd evaluated in int, stored as float,
 * instantiated <u8,128,f32,0,1,1> (leandvb.cc:215). */
void lo_cconverter_u8(const lo_cu8 *in, size_t n, lo_cf32 *out) {
  for (size_t i = 0; i < n; ++i) {
    out[i].re = 0 + (in[i].re - (uint8_t)128) * 1 / 1;
    out[i].im = 0 + (in[i].im - (uint8_t)128) * 1 / 1;
  }
}

/* dsp.h:149-156 with complex*T of math.h:45-48: (re*k, im*k). */
void lo_scaler(float scale, const lo_cf32 *in, size_t n, lo_cf32 *out) {
  for (size_t i = 0; i < n; ++i) {
    out[i].re = in[i].re * scale;
    out[i].im = in[i].im * scale;
  }
}

/* generic.h:256-262 */
size_t lo_decimator(unsigned d, const lo_cf32 *in, size_t n, lo_cf32 *out, size_t cap) {
  size_t count = n / d;
  if (count > cap) count = cap;
  for (size_t m = 0; m < count; ++m) out[m] = in[m * d];
  return count;
}

/* dsp.h:271-280.  `i-ncoeffs/2` is evaluated in *unsigned int* (ncoeffs is
 * `unsigned int`), so for i < ncoeffs/2 the offset wraps to ~2^32 before the
 * conversion to double.  a = (double)(2*M_PI*f) * (double)(unsigned) -> float. */
void lo_fir_shift_coeffs(unsigned ncoeffs, const float *coeffs, float freq, lo_cf32 *shifted) {
  for (int i = 0; i < (int)ncoeffs; ++i) {
    float a = 2 * M_PI * freq * (i - ncoeffs / 2);
    float c = cosf(a), s = sinf(a);
    shifted[i].re = coeffs[i] * c;
    shifted[i].im = coeffs[i] * s;
  }
}

/* dsp.h:233-262.  Output m: x = 0; for i=0..N-1: x = x + sc[i]*in[N + m*D - i],
 * complex*complex per math.h:40-43 with a = coefficient, b = sample. */
size_t lo_fir_filter(unsigned ncoeffs, const lo_cf32 *sc, unsigned decim,
                     const lo_cf32 *in, size_t n_in, lo_cf32 *out, size_t cap,
                     size_t *consumed) {
  *consumed = 0;
  if (n_in < ncoeffs) return 0;
  size_t count = (n_in - ncoeffs) / decim;
  if (count > cap) count = cap;
  for (size_t m = 0; m < count; ++m) {
    const lo_cf32 *pi = in + ncoeffs + m * decim;
    float xr = 0, xi = 0;
    for (unsigned i = 0; i < ncoeffs; ++i, --pi) {
      float pr = sc[i].re * pi->re - sc[i].im * pi->im;
      float pq = sc[i].re * pi->im + sc[i].im * pi->re;
      xr = xr + pr;
      xi = xi + pq;
    }
    out[m].re = xr;
    out[m].im = xi;
  }
  *consumed = count * decim;
  return count;
}

/* The arithmetic of the TOLERANCE modes of the HIP fir_filter (LSDR_FIR_FMA, LSDR_FIR_MFMA), stated once so that the
 * kernels can be pinned to it bit for bit: the reference's loop (dsp.h:246-262) with each product-and-add fused, i.e. what
 * the reference itself computes when built for an FMA machine with -ffp-contract=fast.  fmaf() is a correctly rounded fused
 * multiply-add whatever the host.  Taps with zero imaginary part (current_freq == 0) take the two-chain form (a product
 * with ±0 would be an exact no-op anyway). */
size_t lo_fir_filter_fma(unsigned ncoeffs, const lo_cf32 *sc, unsigned decim,
                         const lo_cf32 *in, size_t n_in, lo_cf32 *out, size_t cap,
                         size_t *consumed) {
  *consumed = 0;
  if (n_in < ncoeffs) return 0;
  size_t count = (n_in - ncoeffs) / decim;
  if (count > cap) count = cap;
  int all_real = 1;
  for (unsigned i = 0; i < ncoeffs; ++i) all_real &= sc[i].im == 0.0f;
  for (size_t m = 0; m < count; ++m) {
    const lo_cf32 *pi = in + ncoeffs + m * decim;
    float xr = 0, xi = 0;
    for (unsigned i = 0; i < ncoeffs; ++i, --pi) {
      xr = fmaf(sc[i].re, pi->re, xr);
      xi = fmaf(sc[i].re, pi->im, xi);
      if (!all_real) {
        xr = fmaf(-sc[i].im, pi->im, xr);
        xi = fmaf(sc[i].im, pi->re, xi);
      }
    }
    out[m].re = xr;
    out[m].im = xi;
  }
  *consumed = count * decim;
  return count;
}

/* The arithmetic of LSDR_FIR_MFMA_BLK (k_fir_mfma_stream / k_fir_mfma_blk), stated once: the reference's loop (dsp.h:246-262) with the
 * taps cut into blocks of `decim` consecutive taps — each block an fmaf chain from zero in tap order, the block sums added in block
 * order with plain float adds.  (What a block-polyphase evaluation on fused hardware computes; same error class as
 * lo_fir_filter_fma.)  Complex taps (current_freq != 0): a block's chain takes its taps in groups of four (one step of the matrix
 * pipe) — the four re-part products of a group, then its four im-part products:
 *    re: fma(cr, xr) x 4, fma(-ci, xi) x 4, next group ...      im: fma(cr, xi) x 4, fma(ci, xr) x 4, next group ... */
size_t lo_fir_filter_blk(unsigned ncoeffs, const lo_cf32 *sc, unsigned decim,
                         const lo_cf32 *in, size_t n_in, lo_cf32 *out, size_t cap,
                         size_t *consumed) {
  *consumed = 0;
  if (n_in < ncoeffs) return 0;
  size_t count = (n_in - ncoeffs) / decim;
  if (count > cap) count = cap;
  int all_real = 1;
  for (unsigned i = 0; i < ncoeffs; ++i) all_real &= sc[i].im == 0.0f;
  for (size_t m = 0; m < count; ++m) {
    const lo_cf32 *p0 = in + ncoeffs + m * decim;
    float yr = 0, yi = 0;
    for (unsigned q = 0; q * decim < ncoeffs; ++q) {
      float zr = 0, zi = 0;
      const unsigned i1 = (q + 1) * decim < ncoeffs ? (q + 1) * decim : ncoeffs;
      for (unsigned g = q * decim; g < i1; g += 4) {
        const unsigned g1 = g + 4 < i1 ? g + 4 : i1;
        for (unsigned i = g; i < g1; ++i) {
          zr = fmaf(sc[i].re, (p0 - i)->re, zr);
          zi = fmaf(sc[i].re, (p0 - i)->im, zi);
        }
        if (!all_real)
          for (unsigned i = g; i < g1; ++i) {
            zr = fmaf(-sc[i].im, (p0 - i)->im, zr);
            zi = fmaf(sc[i].im, (p0 - i)->re, zi);
          }
      }
      yr = q ? yr + zr : zr;
      yi = q ? yi + zi : zi;
    }
    out[m].re = yr;
    out[m].im = yi;
  }
  *consumed = count * decim;
  return count;
}

/* dsp.h:351-360: a = 2*M_PI*f*i with int i. */
void lo_fir_resampler_shift_coeffs(unsigned ncoeffs, const float *coeffs, float freq, lo_cf32 *shifted) {
  for (int i = 0; i < (int)ncoeffs; ++i) {
    float a = 2 * M_PI * freq * i;
    float c = cosf(a), s = sinf(a);
    shifted[i].re = coeffs[i] * c;
    shifted[i].im = coeffs[i] * s;
  }
}

/* dsp.h:306-337 */
size_t lo_fir_resampler(unsigned ncoeffs, const lo_cf32 *sc, int interp,
                        const lo_cf32 *in, size_t n_in, lo_cf32 *out, size_t cap,
                        size_t *consumed) {
  *consumed = 0;
  if (n_in < ncoeffs) return 0;
  if (n_in * interp < ncoeffs) return 0;
  size_t count = (n_in * interp - ncoeffs) / interp;
  if (count > cap / interp) count = cap / interp;
  int latency = (ncoeffs + interp) / interp;
  lo_cf32 *pout = out;
  for (size_t m = 0; m < count; ++m) {
    const lo_cf32 *pin = in + latency + m;
    for (int i = 0; i < interp; ++i, ++pout) {
      const lo_cf32 *pi = pin;
      float xr = 0, xi = 0;
      for (unsigned k = i; k < ncoeffs; k += interp, --pi) {
        float pr = sc[k].re * pi->re - sc[k].im * pi->im;
        float pq = sc[k].re * pi->im + sc[k].im * pi->re;
        xr = xr + pr;
        xi = xi + pq;
      }
      pout->re = xr;
      pout->im = xi;
    }
  }
  *consumed = count;
  return count * interp;
}

/* dsp.h:56-116.  Twiddles: float a = 2.0*M_PI*i/n (double -> float), cosf/sinf;
 * reverse uses conjugate twiddles and a final multiply by (float)(1.0/n). */
void lo_cfft(int n, lo_cf32 *data, int reverse) {
  int logn = 0;
  for (int t = n; t > 1; t >>= 1) ++logn;
  lo_cf32 *om = (lo_cf32 *)malloc(sizeof(lo_cf32) * n);
  for (int i = 0; i < n; ++i) {
    float a = 2.0 * M_PI * i / n;
    om[i].re = cosf(a);
    om[i].im = reverse ? -sinf(a) : sinf(a);
  }
  for (int i = 0; i < n; ++i) {
    int r = 0;
    for (int b = 0; b < logn; ++b) r = (r << 1) | ((i >> b) & 1);
    if (r < i) { lo_cf32 tmp = data[i]; data[i] = data[r]; data[r] = tmp; }
  }
  for (int i = 0; i < logn; ++i) {
    int hbs = 1 << i;
    int dom = 1 << (logn - 1 - i);
    for (int j = 0; j < dom; ++j) {
      int p = j * hbs * 2, q = p + hbs;
      for (int k = 0; k < hbs; ++k) {
        lo_cf32 w = om[k * dom];
        lo_cf32 d = data[q + k];
        float xr = w.re * d.re - w.im * d.im;
        float xi = w.re * d.im + w.im * d.re;
        data[q + k].re = data[p + k].re - xr;
        data[q + k].im = data[p + k].im - xi;
        data[p + k].re = data[p + k].re + xr;
        data[p + k].im = data[p + k].im + xi;
      }
    }
  }
  if (reverse) {
    float invn = 1.0 / n;
    for (int i = 0; i < n; ++i) { data[i].re *= invn; data[i].im *= invn; }
  }
  free(om);
}

/* ---- auto_notch<f32>, sdr.h:46-154 ------------------------------------- */
#define ANF_N 4096
struct lo_auto_notch {
  int nslots, decimation, phase;
  float k, gain, agc_rms_setpoint;
  struct slot { int i; lo_cf32 estim; lo_cf32 *expj; } *slots;
};

/* sdr.h:50-63.  The reference leaves estim/expj uninitialised until the first
 * detect(); the observed behaviour (bit-exact pass-through) is what
 * zero-initialisation gives, and that is the contract here. */
lo_auto_notch *lo_auto_notch_new(int nslots, int decimation, float k, float agc_rms_setpoint) {
  lo_auto_notch *a = (lo_auto_notch *)calloc(1, sizeof(*a));
  a->nslots = nslots; a->decimation = decimation; a->k = k;
  a->gain = 1; a->agc_rms_setpoint = agc_rms_setpoint; a->phase = 0;
  a->slots = (struct slot *)calloc(nslots > 0 ? nslots : 1, sizeof(struct slot));
  for (int s = 0; s < nslots; ++s) {
    a->slots[s].i = -1;
    a->slots[s].expj = (lo_cf32 *)calloc(ANF_N, sizeof(lo_cf32));
  }
  return a;
}
void lo_auto_notch_free(lo_auto_notch *a) {
  for (int s = 0; s < a->nslots; ++s) free(a->slots[s].expj);
  free(a->slots); free(a);
}
int lo_auto_notch_slot_bin(const lo_auto_notch *a, int slot) { return a->slots[slot].i; }

/* sdr.h:76-118 */
static void anf_detect(lo_auto_notch *a, const lo_cf32 *pin) {
  static lo_cf32 data[ANF_N];
  static float amp[ANF_N];
  float m0 = 0, m2 = 0;
  for (int i = 0; i < ANF_N; ++i) {
    data[i] = pin[i];
    m2 += (float)pin[i].re * pin[i].re + (float)pin[i].im * pin[i].im;
    if (fabsf(pin[i].re) > m0) m0 = fabsf(pin[i].re);
    if (fabsf(pin[i].im) > m0) m0 = fabsf(pin[i].im);
  }
  if (a->agc_rms_setpoint && m2) {
    float rms = sqrtf(m2 / ANF_N);
    float new_gain = a->agc_rms_setpoint / rms;
    a->gain = a->gain * 0.9 + new_gain * 0.1; /* double arithmetic, sdr.h:93 */
  }
  lo_cfft(ANF_N, data, 1);
  for (int i = 0; i < ANF_N; ++i) amp[i] = hypotf(data[i].re, data[i].im);
  for (int s = 0; s < a->nslots; ++s) {
    struct slot *sl = &a->slots[s];
    int iamax = 0;
    for (int i = 0; i < ANF_N; ++i)
      if (amp[i] > amp[iamax]) iamax = i;
    if (iamax != sl->i) {
      sl->i = iamax;
      sl->estim.re = 0; sl->estim.im = 0;
      for (int i = 0; i < ANF_N; ++i) {
        float ang = 2 * M_PI * sl->i * i / ANF_N; /* ((2*M_PI)*bin)*i/n in double */
        sl->expj[i].re = cosf(ang);
        sl->expj[i].im = sinf(ang);
      }
    }
    amp[iamax] = 0;
    if (iamax - 1 >= 0) amp[iamax - 1] = 0;
    if (iamax + 1 < ANF_N) amp[iamax + 1] = 0;
  }
}

/* sdr.h:119-138 */
static void anf_process(lo_auto_notch *a, const lo_cf32 *pin, lo_cf32 *pout) {
  const float k = a->k;
  for (int n = 0; n < ANF_N; ++n) {
    float outre = pin[n].re, outim = pin[n].im;
    for (int s = 0; s < a->nslots; ++s) {
      struct slot *sl = &a->slots[s];
      lo_cf32 e = sl->expj[n];
      float bbre = pin[n].re * e.re + pin[n].im * e.im;
      float bbim = -pin[n].re * e.im + pin[n].im * e.re;
      sl->estim.re = bbre * k + sl->estim.re * (1 - k);
      sl->estim.im = bbim * k + sl->estim.im * (1 - k);
      float subre = sl->estim.re * e.re - sl->estim.im * e.im;
      float subim = sl->estim.re * e.im + sl->estim.im * e.re;
      outre -= subre;
      outim -= subim;
    }
    pout[n].re = a->gain * outre;
    pout[n].im = a->gain * outim;
  }
}

/* sdr.h:64-75 */
size_t lo_auto_notch_run(lo_auto_notch *a, const lo_cf32 *in, size_t n, lo_cf32 *out) {
  size_t pos = 0;
  while (n - pos >= ANF_N) {
    a->phase += ANF_N;
    if (a->phase >= a->decimation) {
      a->phase -= a->decimation;
      anf_detect(a, in + pos);
    }
    anf_process(a, in + pos, out + pos);
    pos += ANF_N;
  }
  return pos;
}

/* ---- cnr_fft<f32>, sdr.h:1273-1345 -------------------------------------- */
struct lo_cnr_fft {
  float bandwidth, kavg;
  int nfft, decimation, phase;
  float *avgpower; /* NULL until the first spectrum */
};
lo_cnr_fft *lo_cnr_fft_new(float bandwidth, int nfft, int decimation) {
  lo_cnr_fft *c = (lo_cnr_fft *)calloc(1, sizeof(*c));
  c->bandwidth = bandwidth; c->nfft = nfft; c->decimation = decimation;
  c->kavg = 0.1; c->phase = 0; c->avgpower = NULL;
  return c;
}
void lo_cnr_fft_free(lo_cnr_fft *c) { free(c->avgpower); free(c); }

static float cnr_avgslots(const lo_cnr_fft *c, int i0, int i1) { /* sdr.h:1334-1338 */
  float s = 0;
  for (int i = i0; i <= i1; ++i) s += c->avgpower[i & (c->nfft - 1)];
  return s / (i1 - i0 + 1);
}

/* sdr.h:1291-1332; consumes floor(n/nfft) blocks; returns number of CNR values written */
size_t lo_cnr_fft_run(lo_cnr_fft *c, float freq_tap, float tap_multiplier,
                      const lo_cf32 *in, size_t n, float *out, size_t cap) {
  size_t pos = 0, nout = 0;
  const int N = c->nfft;
  lo_cf32 *data = (lo_cf32 *)malloc(sizeof(lo_cf32) * N);
  float *power = (float *)malloc(sizeof(float) * N);
  while (n - pos >= (size_t)N && nout < cap) {
    c->phase += N;
    if (c->phase >= c->decimation) {
      c->phase -= c->decimation;
      float center_freq = freq_tap * tap_multiplier;
      int icf = floor(center_freq * N + 0.5);
      memcpy(data, in + pos, sizeof(lo_cf32) * N);
      lo_cfft(N, data, 1);
      for (int i = 0; i < N; ++i) power[i] = data[i].re * data[i].re + data[i].im * data[i].im;
      if (!c->avgpower) {
        c->avgpower = (float *)malloc(sizeof(float) * N);
        memcpy(c->avgpower, power, sizeof(float) * N);
      }
      for (int i = 0; i < N; ++i)
        c->avgpower[i] = c->avgpower[i] * (1 - c->kavg) + power[i] * c->kavg;
      int bwslots = (c->bandwidth / 4) * N;
      if (bwslots) {
        float c2plusn2 = cnr_avgslots(c, icf - bwslots, icf + bwslots);
        float n2 = (cnr_avgslots(c, icf - bwslots * 4, icf - bwslots * 3) +
                    cnr_avgslots(c, icf + bwslots * 3, icf + bwslots * 4)) / 2;
        float c2 = c2plusn2 - n2;
        float cnr = (c2 > 0 && n2 > 0) ? 10 * logf(c2 / n2) / logf(10) : -50;
        out[nout++] = cnr;
      }
    }
    pos += N;
  }
  free(data); free(power);
  return nout;
}

/* ---- spectrum<f32>, sdr.h:1347-1404 ---------------------------------------- */
struct lo_spectrum {
  float kavg;
  int decimation, phase;
  float *avgpower; /* NULL until the first spectrum */
};
lo_spectrum *lo_spectrum_new(int decimation, float kavg) {
  lo_spectrum *c = (lo_spectrum *)calloc(1, sizeof(*c));
  c->decimation = decimation; c->kavg = kavg; c->phase = 0; c->avgpower = NULL;
  return c;
}
void lo_spectrum_free(lo_spectrum *c) { free(c->avgpower); free(c); }

/* run(), sdr.h:1361-1370 + do_spectrum(), :1374-1396; out = cap rows of 1024 floats; returns rows written */
size_t lo_spectrum_run(lo_spectrum *c, const lo_cf32 *in, size_t n, float *out, size_t cap) {
  enum { N = 1024 };
  size_t pos = 0, nout = 0;
  lo_cf32 *data = (lo_cf32 *)malloc(sizeof(lo_cf32) * N);
  float power[N];
  while (n - pos >= (size_t)N && nout < cap) {
    c->phase += N;
    if (c->phase >= c->decimation) {
      c->phase -= c->decimation;
      memcpy(data, in + pos, sizeof(lo_cf32) * N);
      lo_cfft(N, data, 1);
      for (int i = 0; i < N; ++i) power[i] = (float)data[i].re * data[i].re + (float)data[i].im * data[i].im;
      if (!c->avgpower) {
        c->avgpower = (float *)malloc(sizeof(float) * N);
        memcpy(c->avgpower, power, sizeof(float) * N);
      }
      for (int i = 0; i < N; ++i) c->avgpower[i] = c->avgpower[i] * (1 - c->kavg) + power[i] * c->kavg;
      float *row = out + nout * N;
      for (int i = 0; i < N / 2; ++i) {        /* dB + fftshift, sdr.h:1390-1393 */
        row[i] = 10 * log10f(c->avgpower[N / 2 + i]);
        row[N / 2 + i] = 10 * log10f(c->avgpower[i]);
      }
      ++nout;
    }
    pos += N;
  }
  free(data);
  return nout;
}

/* ---- rotator<f32>, sdr.h:1226-1259 ------------------------------------------- */
struct lo_rotator { float lut_cos[65536], lut_sin[65536]; unsigned short index; };
lo_rotator *lo_rotator_new(float freq) {
  lo_rotator *r = (lo_rotator *)calloc(1, sizeof(*r));
  int ifreq = freq * 65536;
  for (int i = 0; i < 65536; ++i) {                       /* 2*M_PI * i * ifreq / 65536 in double, cosf of its float value */
    r->lut_cos[i] = cosf(2 * M_PI * i * ifreq / 65536);
    r->lut_sin[i] = sinf(2 * M_PI * i * ifreq / 65536);
  }
  r->index = 0;
  return r;
}
void lo_rotator_free(lo_rotator *r) { free(r); }
void lo_rotator_run(lo_rotator *r, const lo_cf32 *in, size_t n, lo_cf32 *out) {
  for (size_t k = 0; k < n; ++k, ++r->index) {
    const float c = r->lut_cos[r->index], s = r->lut_sin[r->index];
    out[k].re = in[k].re * c - in[k].im * s;
    out[k].im = in[k].re * s + in[k].im * c;
  }
}
