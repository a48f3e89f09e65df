// leansdr_amd/csrc/host_tables.cpp — host-side coefficient and table design.
//
// The kernels take their coefficients / look-up tables as inputs; to reproduce
// the reference bit for bit they must be designed with the same libm calls and
// the same float/double promotions as the reference (SURVEY §7 "hard parts" 4,
// §8 a22).  This file re-derives them from the behaviour of
//   filtergen.h:26-92, math.h:95-111, sdr.h:326-560, dvb.h:45-81, dsp.h:271-280.
// Compiled with -ffp-contract=off.
#include <cmath>
#include "lsdr_internal.h"

namespace {
const float kCstlnAmp = 75.0f;  // sdr.h:297

// C++ promotions of the reference are spelled out with explicit casts here.
inline float f(double x) { return (float)x; }

struct sympt { signed char re, im; };

// sdr.h:489-492: angle = ((i*2) in float) * M_PI / n in double, rounded to float.
sympt polar(float r, int n, float i) {
  float a = f((double)(i * 2.0f) * M_PI / (double)n);
  sympt s;
  s.re = (signed char)(r * cosf(a) * kCstlnAmp);
  s.im = (signed char)(r * sinf(a) * kCstlnAmp);
  return s;
}
// sdr.h:494-501
sympt polar_pi(float r, float frac) {
  float phi = f((double)frac * M_PI);
  sympt s;
  s.re = (signed char)(r * cosf(phi) * kCstlnAmp);
  s.im = (signed char)(r * sinf(phi) * kCstlnAmp);
  return s;
}
}  // namespace

namespace lsdr {

// dsp.h:271-280.  The reference evaluates (i - ncoeffs/2) in unsigned int.
void fir_shift_coeffs(unsigned ncoeffs, const float *coeffs, float freq, lsdr_cf32 *shifted) {
  const double w = 2 * M_PI * (double)freq;
  for (unsigned i = 0; i < ncoeffs; ++i) {
    unsigned off = i - ncoeffs / 2;  // wraps for i < ncoeffs/2, as in the reference
    float a = f(w * (double)off);
    shifted[i].re = coeffs[i] * cosf(a);
    shifted[i].im = coeffs[i] * sinf(a);
  }
}

// sdr.h:529-560
static void fill_lut(cstln_tables &t) {
  t.cost.assign(65536, 0);
  t.phase_error.assign(65536, 0);
  t.symbol.assign(65536, 0);
  for (int I = -128; I < 128; ++I)
    for (int Q = -128; Q < 128; ++Q) {
      int best = 0;
      int32_t d_best = 131072, d_second = 131072;
      for (int s = 0; s < t.nsymbols; ++s) {
        int dI = I - t.symbols[s][0], dQ = Q - t.symbols[s][1];
        int32_t d2 = dI * dI + dQ * dQ;
        if (d2 < d_best) { d_second = d_best; d_best = d2; best = s; }
        else if (d2 < d_second) d_second = d2;
      }
      if (d_best > 32767) d_best = 32767;
      if (d_second > 32767) d_second = 32767;
      size_t idx = (size_t)(I & 255) * 256 + (Q & 255);
      t.cost[idx] = (int16_t)(d_best - d_second);
      t.symbol[idx] = (uint8_t)best;
      float ph_sym = atan2f((float)t.symbols[best][1], (float)t.symbols[best][0]);
      float ph_err = atan2f((float)Q, (float)I) - ph_sym;
      // (s32)(ph_err*65536/(2*M_PI)): float product, double quotient, truncation, mod 2^16
      t.phase_error[idx] = (int16_t)(long)((double)(ph_err * 65536.0f) / (2 * M_PI));
    }
}

// sdr.h:502-527
static void qam(cstln_tables &t, int n) {
  t.nrotations = 4;
  t.nsymbols = n;
  int m = (int)sqrtl((long double)n);
  int q = m / 2;
  float avgpower = f(2 * (q * 0.25 + (q - 1) * q / 2 + (q - 1) * q * (2 * q - 1) / 6) / q);
  float scale = f(1.0 / (double)sqrtf(avgpower));
  int s = 0;
  for (int x = 0; x < m; ++x)
    for (int y = 0; y < m; ++y, ++s) {
      float I = x - (float)(m - 1) / 2, Q = y - (float)(m - 1) / 2;
      t.symbols[s][0] = (signed char)(I * scale * kCstlnAmp);
      t.symbols[s][1] = (signed char)(Q * scale * kCstlnAmp);
    }
}

// make_dvbs2_constellation (dvb.h:45-81) + cstln_lut ctor (sdr.h:326-468)
// fast_qpsk_receiver::init_lookup_tables, sdr.h:1154-1171.  Same libm calls and conversions as the reference:
// atan2f(float,float)·65536 in float, /(2π) in double, → s_angle (the +π entries, 32768.0…, wrap to −32768 like the
// reference build: via int32); hypotf → int; 128 + 75·cosf(float(2πa/65536)) → u8; (int)(128 + r·cos(2πa/256)) → u8.
void build_fastqpsk_tables(unsigned *polar, unsigned short *rect, unsigned short *sincos) {
  for (int i = 0; i < 256; ++i)
    for (int q = 0; q < 256; ++q) {
      const double v = atan2f((float)(q - 128), (float)(i - 128)) * 65536 / (2 * M_PI);
      const unsigned a = (unsigned)(uint16_t)(int16_t)(int32_t)v;
      const unsigned r = (unsigned)(uint8_t)(int)hypotf((float)(i - 128), (float)(q - 128));
      polar[i * 256 + q] = a | (r << 16);
    }
  for (unsigned long a = 0; a < 65536; ++a) {
    const float f = 2 * M_PI * a / 65536;
    const unsigned re = (uint8_t)(128 + 75.0f * cosf(f)), im = (uint8_t)(128 + 75.0f * sinf(f));
    sincos[a] = (unsigned short)(re | (im << 8));
  }
  for (int a = 0; a < 256; ++a)
    for (int r = 0; r < 256; ++r) {
      const unsigned re = (uint8_t)(int)(128 + r * cos(2 * M_PI * a / 256)), im = (uint8_t)(int)(128 + r * sin(2 * M_PI * a / 256));
      rect[a * 256 + r] = (unsigned short)(re | (im << 8));
    }
}

int build_cstln(int predef, int fec, cstln_tables &t) {
  float g1 = 1, g2 = 1, g3 = 1;
  if (predef == LSDR_APSK16) {
    switch (fec) {
      case LSDR_FEC23: case LSDR_FEC46: g1 = f(3.15); break;
      case LSDR_FEC34: g1 = f(2.85); break;
      case LSDR_FEC45: g1 = f(2.75); break;
      case LSDR_FEC56: g1 = f(2.70); break;
      case LSDR_FEC89: g1 = f(2.60); break;
      case LSDR_FEC910: g1 = f(2.57); break;
      default: return LSDR_E_ARG;
    }
  } else if (predef == LSDR_APSK32) {
    switch (fec) {
      case LSDR_FEC34: g1 = f(2.84); g2 = f(5.27); break;
      case LSDR_FEC45: g1 = f(2.72); g2 = f(4.87); break;
      case LSDR_FEC56: g1 = f(2.64); g2 = f(4.64); break;
      case LSDR_FEC89: g1 = f(2.54); g2 = f(4.33); break;
      case LSDR_FEC910: g1 = f(2.53); g2 = f(4.30); break;
      default: return LSDR_E_ARG;
    }
  } else if (predef == LSDR_APSK64E) {
    g1 = f(2.4); g2 = f(4.3); g3 = 7;
  }
  memset(t.symbols, 0, sizeof(t.symbols));
  auto put = [&](int k, sympt s) { t.symbols[k][0] = s.re; t.symbols[k][1] = s.im; };
  switch (predef) {
    case LSDR_BPSK:
      t.nrotations = 2; t.nsymbols = 2;
      put(0, polar(1, 8, 1)); put(1, polar(1, 8, 5));
      break;
    case LSDR_QPSK: {
      t.nrotations = 4; t.nsymbols = 4;
      const float pos[4] = {0.5f, 3.5f, 1.5f, 2.5f};
      for (int k = 0; k < 4; ++k) put(k, polar(1, 4, pos[k]));
      break;
    }
    case LSDR_PSK8: {
      t.nrotations = 8; t.nsymbols = 8;
      const float pos[8] = {1, 0, 4, 5, 2, 7, 3, 6};
      for (int k = 0; k < 8; ++k) put(k, polar(1, 8, pos[k]));
      break;
    }
    case LSDR_APSK16: {
      float r1 = sqrtf(4 / (1 + 3 * g1 * g1)), r2 = g1 * r1;
      t.nrotations = 4; t.nsymbols = 16;
      const float outer[12] = {1.5f, 10.5f, 4.5f, 7.5f, 0.5f, 11.5f, 5.5f, 6.5f, 2.5f, 9.5f, 3.5f, 8.5f};
      const float inner[4] = {0.5f, 3.5f, 1.5f, 2.5f};
      for (int k = 0; k < 12; ++k) put(k, polar(r2, 12, outer[k]));
      for (int k = 0; k < 4; ++k) put(12 + k, polar(r1, 4, inner[k]));
      break;
    }
    case LSDR_APSK32: {
      float r1 = sqrtf(8 / (1 + 3 * g1 * g1 + 4 * g2 * g2)), r2 = g1 * r1, r3 = g2 * r1;
      t.nrotations = 4; t.nsymbols = 32;
      const float mid[8] = {1.5f, 2.5f, 10.5f, 9.5f, 4.5f, 3.5f, 7.5f, 8.5f};
      const float outA[8] = {1, 3, 14, 12, 6, 4, 9, 11};
      const float outB[8] = {0, 2, 15, 13, 7, 5, 8, 10};
      const float mix12[4] = {0.5f, 11.5f, 5.5f, 6.5f};
      const float mix4[4] = {0.5f, 3.5f, 1.5f, 2.5f};
      for (int k = 0; k < 8; ++k) put(k, polar(r2, 12, mid[k]));
      for (int k = 0; k < 8; ++k) put(8 + k, polar(r3, 16, outA[k]));
      for (int k = 0; k < 4; ++k) {
        put(16 + 2 * k, polar(r2, 12, mix12[k]));
        put(17 + 2 * k, polar(r1, 4, mix4[k]));
      }
      for (int k = 0; k < 8; ++k) put(24 + k, polar(r3, 16, outB[k]));
      break;
    }
    case LSDR_APSK64E: {
      float r1 = sqrtf(64 / (4 + 12 * g1 * g1 + 20 * g2 * g2 + 28 * g3 * g3));
      float r2 = g1 * r1, r3 = g2 * r1, r4 = g3 * r1;
      t.nrotations = 4; t.nsymbols = 64;
      // EN 302 307-2 Table 13e as laid out at sdr.h:439-454: {radius, 4 angles in units of pi}
      struct row { int ring; double a[4]; };
      const row rows[16] = {
          {4, {1.0 / 4, 7.0 / 4, 3.0 / 4, 5.0 / 4}},     {4, {13.0 / 28, 43.0 / 28, 15.0 / 28, 41.0 / 28}},
          {4, {1.0 / 28, 55.0 / 28, 27.0 / 28, 29.0 / 28}}, {1, {1.0 / 4, 7.0 / 4, 3.0 / 4, 5.0 / 4}},
          {4, {9.0 / 28, 47.0 / 28, 19.0 / 28, 37.0 / 28}}, {4, {11.0 / 28, 45.0 / 28, 17.0 / 28, 39.0 / 28}},
          {3, {1.0 / 20, 39.0 / 20, 19.0 / 20, 21.0 / 20}}, {2, {1.0 / 12, 23.0 / 12, 11.0 / 12, 13.0 / 12}},
          {4, {5.0 / 28, 51.0 / 28, 23.0 / 28, 33.0 / 28}}, {3, {9.0 / 20, 31.0 / 20, 11.0 / 20, 29.0 / 20}},
          {4, {3.0 / 28, 53.0 / 28, 25.0 / 28, 31.0 / 28}}, {2, {5.0 / 12, 19.0 / 12, 7.0 / 12, 17.0 / 12}},
          {3, {1.0 / 4, 7.0 / 4, 3.0 / 4, 5.0 / 4}},     {3, {7.0 / 20, 33.0 / 20, 13.0 / 20, 27.0 / 20}},
          {3, {3.0 / 20, 37.0 / 20, 17.0 / 20, 23.0 / 20}}, {2, {1.0 / 4, 7.0 / 4, 3.0 / 4, 5.0 / 4}}};
      const float radius[5] = {0, r1, r2, r3, r4};
      for (int b = 0; b < 16; ++b)
        for (int j = 0; j < 4; ++j) put(4 * b + j, polar_pi(radius[rows[b].ring], f(rows[b].a[j])));
      break;
    }
    case LSDR_QAM16: qam(t, 16); break;
    case LSDR_QAM64: qam(t, 64); break;
    case LSDR_QAM256: qam(t, 256); break;
    default: return LSDR_E_ARG;
  }
  fill_lut(t);
  return t.nsymbols;
}

}  // namespace lsdr

extern "C" {

void lsdr_filtergen_normalize_dcgain(int n, float *c, float gain) {
  float s = 0;
  for (int i = 0; i < n; ++i) s = s + c[i];
  if (s) gain /= s;
  for (int i = 0; i < n; ++i) c[i] = c[i] * gain;
}

void lsdr_filtergen_normalize_power(int n, float *c, float gain) {
  float s2 = 0;
  for (int i = 0; i < n; ++i) s2 = s2 + c[i] * c[i];
  if (s2) gain /= sqrtf(s2);
  for (int i = 0; i < n; ++i) c[i] = c[i] * gain;
}

// filtergen.h:45-62: windowed-sinc (rectangular) low-pass, DC gain normalised.
int lsdr_filtergen_lowpass(int order, float Fcut, float gain, float *c) {
  if (order < 0 || !c) return LSDR_E_ARG;
  const int n = order + 1;
  for (int i = 0; i < n; ++i) {
    float t = f(i - (n - 1) * 0.5);
    double x = 2 * M_PI * (double)Fcut * (double)t;
    c[i] = f((double)(2 * Fcut) * (t ? sin(x) / x : 1.0));
  }
  lsdr_filtergen_normalize_dcgain(n, c, gain);
  return n;
}

// filtergen.h:68-92: root-raised-cosine; single-precision libm throughout
// (the reference's float arguments select the float overloads).
int lsdr_filtergen_root_raised_cosine(int order, float Fs, float rolloff, float *c) {
  if (order < 0 || !c) return LSDR_E_ARG;
  const float B = rolloff, pi = f(M_PI);
  const int n = (order + 1) | 1;
  for (int i = 0; i < n; ++i) {
    int t = i - n / 2;
    float v;
    if (t == 0) {
      v = sqrtf(Fs) * (1 - B + 4 * B / pi);
    } else {
      float tT = t * Fs;
      float den = pi * tT * (1 - (4 * B * tT) * (4 * B * tT));
      if (!den)
        v = B * sqrtf(Fs / 2) * ((1 + 2 / pi) * sinf(pi / (4 * B)) + (1 - 2 / pi) * cosf(pi / (4 * B)));
      else
        v = sqrtf(Fs) * (sinf(pi * tT * (1 - B)) + 4 * B * tT * cosf(pi * tT * (1 + B))) / den;
    }
    c[i] = v;
  }
  lsdr_filtergen_normalize_dcgain(n, c, 1);
  return n;
}

// math.h:98-104
void lsdr_trig16_table(lsdr_cf32 *lut) {
  for (int a = 0; a < 65536; ++a) {
    float af = f((double)(a * 2) * M_PI / 65536);
    lut[a].re = cosf(af);
    lut[a].im = sinf(af);
  }
}

int lsdr_cstln_lut_build(int predef, int fec, int16_t *cost, uint8_t *symbol, int16_t *phase_error,
                         int8_t *symbols, int *nrotations) {
  lsdr::cstln_tables t;
  int n = lsdr::build_cstln(predef, fec, t);
  if (n < 0) { lsdr_set_error("unsupported constellation/code rate %d/%d", predef, fec); return n; }
  if (cost) memcpy(cost, t.cost.data(), 65536 * sizeof(int16_t));
  if (symbol) memcpy(symbol, t.symbol.data(), 65536);
  if (phase_error) memcpy(phase_error, t.phase_error.data(), 65536 * sizeof(int16_t));
  if (symbols) memcpy(symbols, t.symbols, 2 * n);
  if (nrotations) *nrotations = t.nrotations;
  return n;
}

}  // extern "C"
