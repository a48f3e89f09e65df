/* oracle/lsdr_oracle_chan.c — CPU ORACLE (test infrastructure, see lsdr_oracle.h) for the channel simulator of
 * leanchansim (leanchansim.cc:34-190): wgn_c<f32> (dsp.h:164-190), adder<cf32> (dsp.h:118-138), drifter<float>
 * (leanchansim.cc:34-88), cconverter<f32,0,u8,128,1,1> (dsp.h:33-54).
 *
 * wgn_c draws from glibc's drand48() and calls glibc's logf(): both are restated here from their published algorithms so
 * that the GPU kernels can be checked draw by draw —
 *   drand48 (glibc 2.35 stdlib/drand48-iter.c, erand48_r.c): X' = (0x5DEECE66D·X + 0xB) mod 2^48, X starts at 0 in a process that never seeds (glibc keeps its state in zeroed static storage),
 *     srand48(s): X = (s<<16)|0x330E; the result is X'·2^-48 (exact in a double);
 *   logf (glibc 2.35 sysdeps/ieee754/flt-32/e_logf.c + e_logf_data.c, from ARM optimized-routines): 16-entry (1/c, log c)
 *     table, degree-3 polynomial evaluated in double, rounded once to float.
 * tests/test_oracle_chan.py pins lo_drand48 against drand48() and lo_logf against logf() of this machine's libm (a sample in
 * the default suite; every float in (0,1) — the whole domain wgn_c uses — with LSDR_EXHAUSTIVE=1: 0 mismatches, with or
 * without FMA contraction of the double expressions). */
#include "lsdr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

uint64_t lo_drand48_default(void) { return 0; }   /* glibc: the static drand48_data starts zeroed (not 0x1234ABCD330E) */
uint64_t lo_srand48(long seed) { return (((uint64_t)(uint32_t)seed) << 16) | 0x330E; }
double lo_drand48(uint64_t *x) {
  *x = (*x * 0x5DEECE66DULL + 0xB) & 0xFFFFFFFFFFFFULL;
  return (double)*x * 0x1p-48;
}

static const double logf_tab[16][2] = {
    {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
    {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},  {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
    {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
    {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
    {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
    {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
    {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3},
    {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2},
};
void lo_logf_table(double *tab32, double *ln2, double *poly3) {
  memcpy(tab32, logf_tab, sizeof(logf_tab));
  *ln2 = 0x1.62e42fefa39efp-1;
  poly3[0] = -0x1.00ea348b88334p-2; poly3[1] = 0x1.5575b0be00b6ap-2; poly3[2] = -0x1.ffffef20a4123p-2;
}
/* positive normal x only (wgn_c calls it on 0 < r2 < 1, never subnormal: |x|,|y| ≥ 2^-47) */
float lo_logf(float x) {
  uint32_t ix, iz;
  memcpy(&ix, &x, 4);
  if (ix == 0x3f800000u) return 0;
  uint32_t tmp = ix - 0x3f330000u;
  int i = (tmp >> 19) % 16, k = (int32_t)tmp >> 23;
  iz = ix - (tmp & (0x1ffu << 23));
  float zf;
  memcpy(&zf, &iz, 4);
  double z = (double)zf, r = z * logf_tab[i][0] - 1, y0 = logf_tab[i][1] + (double)k * 0x1.62e42fefa39efp-1, r2 = r * r;
  double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
  y = -0x1.00ea348b88334p-2 * r2 + y;
  y = y * r2 + (y0 + r);
  return (float)y;
}

/* wgn_c<f32>::run (dsp.h:169-186): n samples; *x is the drand48 state */
void lo_wgn(uint64_t *state, float stddev, lo_cf32 *out, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    float x, y, r2;
    do {
      x = 2 * lo_drand48(state) - 1;
      y = 2 * lo_drand48(state) - 1;
      r2 = x * x + y * y;
    } while (r2 == 0 || r2 >= 1);
    float k = sqrtf(-lo_logf(r2) / r2) * stddev;
    out[i].re = k * x;
    out[i].im = k * y;
  }
}

/* leanchansim.cc:262 / leandvbtx.cc:288: dB option → linear amplitude, through libm like the reference's main() */
float lo_db_to_amp(double db) { return expf(logf(10) * db / 20); }

void lo_adder(const lo_cf32 *a, const lo_cf32 *b, size_t n, lo_cf32 *out) {   /* dsp.h:125-134 */
  for (size_t i = 0; i < n; ++i) { out[i].re = a[i].re + b[i].re; out[i].im = a[i].im + b[i].im; }
}

/* float → integer conversions as x86-64 does them (cvttss2si / cvttsd2si: "integer indefinite" when out of range) */
static int32_t x86_f2i(float v) { return (v >= -2147483648.0f && v < 2147483648.0f) ? (int32_t)v : INT32_MIN; }
static int64_t x86_d2l(double v) { return (v >= -9223372036854775808.0 && v < 9223372036854775808.0) ? (int64_t)v : INT64_MIN; }

void lo_cconv_f32_u8(const lo_cf32 *in, size_t n, lo_cu8 *out) {   /* dsp.h:44-47 with Zin=0, Zout=128, Gn=Gd=1 */
  for (size_t i = 0; i < n; ++i) {
    out[i].re = (uint8_t)x86_f2i(128 + (in[i].re - (float)0) * 1 / 1);
    out[i].im = (uint8_t)x86_f2i(128 + (in[i].im - (float)0) * 1 / 1);
  }
}

void lo_cconv_f32_s16(const lo_cf32 *in, size_t n, int16_t *out /* [n][2] */) {   /* cconverter<f32,0,int16_t,0,32768,1> (leandvbtx.cc:179), dsp.h:44-47 */
  for (size_t i = 0; i < n; ++i) {
    out[2 * i] = (int16_t)x86_f2i(0 + (in[i].re - (float)0) * 32768 / 1);
    out[2 * i + 1] = (int16_t)x86_f2i(0 + (in[i].im - (float)0) * 32768 / 1);
  }
}

/* drifter<float> (leanchansim.cc:34-88).  One call = one run(): `phase` restarts at 0 (it is a local of run()). */
void lo_drifter_trig(lo_cf32 *lut65536) {   /* leanchansim.cc:42-46 */
  for (int i = 0; i < 65536; ++i) {
    float a = 2 * M_PI * i / 65536;
    lut65536[i].re = cosf(a);
    lut65536[i].im = sinf(a);
  }
}
void lo_drifter_run(const lo_cf32 *lut, const float amp[3], const float freq[3], long a[3], const lo_cf32 *in, size_t n, lo_cf32 *out) {
  int16_t phase = 0;
  for (size_t s = 0; s < n; ++s) {
    float f = 0;
    for (int i = 0; i < 3; ++i) {
      const lo_cf32 *r = &lut[(uint16_t)(a[i] >> 16)];
      f += amp[i] * r->im;
      a[i] = x86_d2l((double)a[i] + freq[i] * 4294967296.0);
    }
    phase = (int16_t)x86_f2i((float)phase + f * 65536);
    const lo_cf32 *r = &lut[(uint16_t)phase];
    out[s].re = in[s].re * r->re - in[s].im * r->im;
    out[s].im = in[s].re * r->im + in[s].im * r->re;
  }
}
