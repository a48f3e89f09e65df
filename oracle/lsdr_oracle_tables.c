/* oracle/lsdr_oracle_tables.c — CPU ORACLE (test infrastructure).
 * Tables and coefficient design: trig16, cstln_lut<256>, filtergen.
 * Float/double mixing follows the reference expression by expression; the
 * comments give the C++ promotions that decide the rounding. */
#include "lsdr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* math.h:98-104: `float af = a * 2*M_PI / 65536` -> (a*2) int, *M_PI and /65536 in
 * double, rounded to float; then cosf/sinf. */
void lo_trig16(lo_cf32 *lut) {
  for (int a = 0; a < 65536; ++a) {
    float af = a * 2 * M_PI / 65536;
    lut[a].re = cosf(af);
    lut[a].im = sinf(af);
  }
}

/* math.h:108-110: (uint16_t)(int16_t)(int32_t)a — truncate toward zero, wrap. */
unsigned lo_trig16_index(float a) { return (uint16_t)(int16_t)(int32_t)a; }

static const float cstln_amp = 75; /* sdr.h:297 */

/* sdr.h:489-492: float a = i*2*M_PI/n  ((i*2) float, then double), components
 * r*cosf(a)*cstln_amp in float, converted to signed char (truncation). */
static void polar(int8_t *dst, float r, int n, float i) {
  float a = i * 2 * M_PI / n;
  dst[0] = (signed char)(r * cosf(a) * cstln_amp);
  dst[1] = (signed char)(r * sinf(a) * cstln_amp);
}

/* sdr.h:494-501: phi = a[j]*M_PI (double) rounded to float. */
static void polar2(lo_cstln_lut *c, int i, float r, float a0, float a1, float a2, float a3) {
  float a[4] = {a0, a1, a2, a3};
  for (int j = 0; j < 4; ++j) {
    float phi = a[j] * M_PI;
    c->symbols[i + j][0] = (signed char)(r * cosf(phi) * cstln_amp);
    c->symbols[i + j][1] = (signed char)(r * sinf(phi) * cstln_amp);
  }
}

/* sdr.h:529-560 */
static void make_lut_from_symbols(lo_cstln_lut *c) {
  const int R = 256;
  for (int I = -R / 2; I < R / 2; ++I)
    for (int Q = -R / 2; Q < R / 2; ++Q) {
      int idx = (I & (R - 1)) * 256 + (Q & (R - 1));
      uint8_t nearest = 0;
      int32_t cost = R * R * 2, cost2 = R * R * 2;
      for (int s = 0; s < c->nsymbols; ++s) {
        int32_t d2 = (I - c->symbols[s][0]) * (I - c->symbols[s][0]) +
                     (Q - c->symbols[s][1]) * (Q - c->symbols[s][1]);
        if (d2 < cost) { cost2 = cost; cost = d2; nearest = s; }
        else if (d2 < cost2) { cost2 = d2; }
      }
      if (cost > 32767) cost = 32767;
      if (cost2 > 32767) cost2 = 32767;
      c->cost[idx] = (int16_t)(cost - cost2);
      c->symbol[idx] = nearest;
      float ph_symbol = atan2f(c->symbols[nearest][1], c->symbols[nearest][0]);
      float ph_err = atan2f(Q, I) - ph_symbol;
      /* (s32)(ph_err * 65536 / (2*M_PI)): float*int in float, then double
       * division, truncation to long, stored modulo 2^16 (s_angle). */
      c->phase_error[idx] = (int16_t)(long)(ph_err * 65536 / (2 * M_PI));
    }
}

/* sdr.h:502-527 */
static void make_qam(lo_cstln_lut *c, int n) {
  c->nrotations = 4;
  c->nsymbols = n;
  int m = sqrtl(n);
  float scale;
  {
    int q = m / 2;
    /* int arithmetic except q*0.25 (double): 2*(q*0.25+(q-1)*q/2+(q-1)*q*(2*q-1)/6)/q */
    float avgpower = 2 * (q * 0.25 + (q - 1) * q / 2 + (q - 1) * q * (2 * q - 1) / 6) / q;
    scale = 1.0 / sqrtf(avgpower);
  }
  int s = 0;
  for (int x = 0; x < m; ++x)
    for (int y = 0; y < m; ++y) {
      float I = x - (float)(m - 1) / 2;
      float Q = y - (float)(m - 1) / 2;
      c->symbols[s][0] = (signed char)(I * scale * cstln_amp);
      c->symbols[s][1] = (signed char)(Q * scale * cstln_amp);
      ++s;
    }
  make_lut_from_symbols(c);
}

/* sdr.h:326-468 */
int lo_cstln_lut_init(lo_cstln_lut *c, int predef, float gamma1, float gamma2, float gamma3) {
  memset(c, 0, sizeof(*c));
#define SYM(k, r, n, i) polar(c->symbols[k], r, n, i)
  switch (predef) {
    case LO_BPSK:
      c->nrotations = 2; c->nsymbols = 2;
      SYM(0, 1, 8, 1); SYM(1, 1, 8, 5);
      make_lut_from_symbols(c);
      break;
    case LO_QPSK:
      c->nrotations = 4; c->nsymbols = 4;
      SYM(0, 1, 4, 0.5); SYM(1, 1, 4, 3.5); SYM(2, 1, 4, 1.5); SYM(3, 1, 4, 2.5);
      make_lut_from_symbols(c);
      break;
    case LO_PSK8:
      c->nrotations = 8; c->nsymbols = 8;
      SYM(0, 1, 8, 1); SYM(1, 1, 8, 0); SYM(2, 1, 8, 4); SYM(3, 1, 8, 5);
      SYM(4, 1, 8, 2); SYM(5, 1, 8, 7); SYM(6, 1, 8, 3); SYM(7, 1, 8, 6);
      make_lut_from_symbols(c);
      break;
    case LO_APSK16: {
      float r1 = sqrtf(4 / (1 + 3 * gamma1 * gamma1));
      float r2 = gamma1 * r1;
      c->nrotations = 4; c->nsymbols = 16;
      static const float a12[12] = {1.5, 10.5, 4.5, 7.5, 0.5, 11.5, 5.5, 6.5, 2.5, 9.5, 3.5, 8.5};
      for (int k = 0; k < 12; ++k) SYM(k, r2, 12, a12[k]);
      SYM(12, r1, 4, 0.5); SYM(13, r1, 4, 3.5); SYM(14, r1, 4, 1.5); SYM(15, r1, 4, 2.5);
      make_lut_from_symbols(c);
      break;
    }
    case LO_APSK32: {
      float r1 = sqrtf(8 / (1 + 3 * gamma1 * gamma1 + 4 * gamma2 * gamma2));
      float r2 = gamma1 * r1;
      float r3 = gamma2 * r1;
      c->nrotations = 4; c->nsymbols = 32;
      static const float a0[8] = {1.5, 2.5, 10.5, 9.5, 4.5, 3.5, 7.5, 8.5};
      static const float a8[8] = {1, 3, 14, 12, 6, 4, 9, 11};
      static const float a24[8] = {0, 2, 15, 13, 7, 5, 8, 10};
      for (int k = 0; k < 8; ++k) SYM(k, r2, 12, a0[k]);
      for (int k = 0; k < 8; ++k) SYM(8 + k, r3, 16, a8[k]);
      SYM(16, r2, 12, 0.5); SYM(17, r1, 4, 0.5); SYM(18, r2, 12, 11.5); SYM(19, r1, 4, 3.5);
      SYM(20, r2, 12, 5.5); SYM(21, r1, 4, 1.5); SYM(22, r2, 12, 6.5); SYM(23, r1, 4, 2.5);
      for (int k = 0; k < 8; ++k) SYM(24 + k, r3, 16, a24[k]);
      make_lut_from_symbols(c);
      break;
    }
    case LO_APSK64E: {
      float r1 = sqrtf(64 / (4 + 12 * gamma1 * gamma1 + 20 * gamma2 * gamma2 + 28 * gamma3 * gamma3));
      float r2 = gamma1 * r1, r3 = gamma2 * r1, r4 = gamma3 * r1;
      c->nrotations = 4; c->nsymbols = 64;
      polar2(c, 0, r4, 1.0 / 4, 7.0 / 4, 3.0 / 4, 5.0 / 4);
      polar2(c, 4, r4, 13.0 / 28, 43.0 / 28, 15.0 / 28, 41.0 / 28);
      polar2(c, 8, r4, 1.0 / 28, 55.0 / 28, 27.0 / 28, 29.0 / 28);
      polar2(c, 12, r1, 1.0 / 4, 7.0 / 4, 3.0 / 4, 5.0 / 4);
      polar2(c, 16, r4, 9.0 / 28, 47.0 / 28, 19.0 / 28, 37.0 / 28);
      polar2(c, 20, r4, 11.0 / 28, 45.0 / 28, 17.0 / 28, 39.0 / 28);
      polar2(c, 24, r3, 1.0 / 20, 39.0 / 20, 19.0 / 20, 21.0 / 20);
      polar2(c, 28, r2, 1.0 / 12, 23.0 / 12, 11.0 / 12, 13.0 / 12);
      polar2(c, 32, r4, 5.0 / 28, 51.0 / 28, 23.0 / 28, 33.0 / 28);
      polar2(c, 36, r3, 9.0 / 20, 31.0 / 20, 11.0 / 20, 29.0 / 20);
      polar2(c, 40, r4, 3.0 / 28, 53.0 / 28, 25.0 / 28, 31.0 / 28);
      polar2(c, 44, r2, 5.0 / 12, 19.0 / 12, 7.0 / 12, 17.0 / 12);
      polar2(c, 48, r3, 1.0 / 4, 7.0 / 4, 3.0 / 4, 5.0 / 4);
      polar2(c, 52, r3, 7.0 / 20, 33.0 / 20, 13.0 / 20, 27.0 / 20);
      polar2(c, 56, r3, 3.0 / 20, 37.0 / 20, 17.0 / 20, 23.0 / 20);
      polar2(c, 60, r2, 1.0 / 4, 7.0 / 4, 3.0 / 4, 5.0 / 4);
      make_lut_from_symbols(c);
      break;
    }
    case LO_QAM16: make_qam(c, 16); break;
    case LO_QAM64: make_qam(c, 64); break;
    case LO_QAM256: make_qam(c, 256); break;
    default: return -1;
  }
#undef SYM
  return c->nsymbols;
}

/* dvb.h:45-81 — gamma literals are double constants assigned to float. */
int lo_make_dvbs2_constellation(lo_cstln_lut *c, int predef, int fec) {
  float g1 = 1, g2 = 1, g3 = 1;
  if (predef == LO_APSK16) {
    switch (fec) {
      case LO_FEC23: case LO_FEC46: g1 = 3.15; break;
      case LO_FEC34: g1 = 2.85; break;
      case LO_FEC45: g1 = 2.75; break;
      case LO_FEC56: g1 = 2.70; break;
      case LO_FEC89: g1 = 2.60; break;
      case LO_FEC910: g1 = 2.57; break;
      default: return -1;
    }
  } else if (predef == LO_APSK32) {
    switch (fec) {
      case LO_FEC34: g1 = 2.84; g2 = 5.27; break;
      case LO_FEC45: g1 = 2.72; g2 = 4.87; break;
      case LO_FEC56: g1 = 2.64; g2 = 4.64; break;
      case LO_FEC89: g1 = 2.54; g2 = 4.33; break;
      case LO_FEC910: g1 = 2.53; g2 = 4.30; break;
      default: return -1;
    }
  } else if (predef == LO_APSK64E) {
    g1 = 2.4; g2 = 4.3; g3 = 7;
  }
  return lo_cstln_lut_init(c, predef, g1, g2, g3);
}

/* sdr.h:564-571 */
void lo_cstln_harden(lo_cstln_lut *c) {
  for (int i = 0; i < 65536; ++i) {
    if (c->cost[i] < 0) c->cost[i] = -1;
    if (c->cost[i] > 0) c->cost[i] = 1;
  }
}

/* sdr.h:470-482: halve until inside [-128,127]^2, then [(u8)(s8)I][(u8)(s8)Q]. */
unsigned lo_cstln_lookup_index(float I, float Q) {
  while (I < -128 || I > 127 || Q < -128 || Q > 127) {
    I *= 0.5;
    Q *= 0.5;
  }
  return (unsigned)(uint8_t)(int8_t)I * 256 + (uint8_t)(int8_t)Q;
}

/* filtergen.h:34-40 (all float) */
void lo_normalize_dcgain(int n, float *c, float gain) {
  float s = 0;
  for (int i = 0; i < n; ++i) s = s + c[i];
  if (s) gain /= s;
  for (int i = 0; i < n; ++i) c[i] = c[i] * gain;
}

/* filtergen.h:26-32 */
void lo_normalize_power(int n, float *c, float gain) {
  float s2 = 0;
  for (int i = 0; i < n; ++i) s2 = s2 + c[i] * c[i];
  if (s2) gain /= sqrtf(s2);
  for (int i = 0; i < n; ++i) c[i] = c[i] * gain;
}

/* filtergen.h:45-62.  t = i-(ncoeffs-1)*0.5 (double -> float);
 * sinc = 2*Fcut (float) * (double sin(...)/(...)) -> float.  Rectangular window. */
int lo_lowpass(int order, float Fcut, float *coeffs, float gain) {
  int ncoeffs = order + 1;
  for (int i = 0; i < ncoeffs; ++i) {
    float t = i - (ncoeffs - 1) * 0.5;
    float sinc = 2 * Fcut * (t ? sin(2 * M_PI * Fcut * t) / (2 * M_PI * Fcut * t) : 1);
    float window = 1;
    coeffs[i] = sinc * window;
  }
  lo_normalize_dcgain(ncoeffs, coeffs, gain);
  return ncoeffs;
}

/* filtergen.h:68-92.  In the reference (C++, <math.h> overloads) every sqrt/sin/cos
 * here takes a float argument and therefore resolves to the float version. */
int lo_root_raised_cosine(int order, float Fs, float rolloff, float *coeffs) {
  float B = rolloff, pi = M_PI;
  int ncoeffs = (order + 1) | 1;
  for (int i = 0; i < ncoeffs; ++i) {
    int t = i - ncoeffs / 2;
    float c;
    if (t == 0)
      c = sqrtf(Fs) * (1 - B + 4 * B / pi);
    else {
      float tT = t * Fs;
      float den = pi * tT * (1 - (4 * B * tT) * (4 * B * tT));
      if (!den)
        c = B * sqrtf(Fs / 2) * ((1 + 2 / pi) * sinf(pi / (4 * B)) + (1 - 2 / pi) * cosf(pi / (4 * B)));
      else
        c = sqrtf(Fs) * (sinf(pi * tT * (1 - B)) + 4 * B * tT * cosf(pi * tT * (1 + B))) / den;
    }
    coeffs[i] = c;
  }
  lo_normalize_dcgain(ncoeffs, coeffs, 1);
  return ncoeffs;
}
