/* oracle/lsdr_oracle_tx.c — CPU ORACLE (test infrastructure, see lsdr_oracle.h) for the transmit chain of leandvbtx
 * (leandvbtx.cc:79-175): randomizer (dvb.h:1063-1102), rs_encoder (dvb.h:957-980, rs.h:141-167 in lsdr_oracle_fec.c),
 * interleaver (dvb.h:899-921), dvb_convol on convol_multipoly (dvb.h:567-604, convolutional.h:226-270),
 * cstln_transmitter (sdr.h:1196-1222), simple_agc (sdr.h:238-274).  fir_resampler is in lsdr_oracle_dsp.c. */
#include "lsdr_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* randomizer: XOR with the 8-packet PRBS pattern (byte 0 = 0xff, other sync positions 0), dvb.h:1074-1099 */
size_t lo_randomizer(unsigned *pos /* in/out: 0..1503, multiple of 188 */, const uint8_t *in, size_t npackets, uint8_t *out) {
  uint8_t pattern[1504];
  lo_derandomizer_pattern(pattern);   /* the same precompute_pattern() as the derandomizer's (dvb.h:1116-1129) */
  for (size_t p = 0; p < npackets; ++p) {
    for (int i = 0; i < 188; ++i) out[p * 188 + i] = in[p * 188 + i] ^ pattern[*pos + i];
    *pos += 188;
    if (*pos == 1504) *pos = 0;
  }
  return npackets;
}

/* interleaver::run (dvb.h:905-917): needs 12 packets readable, consumes 1, writes 204 bytes */
size_t lo_interleaver(const uint8_t *in_packets, size_t npackets, uint8_t *out, size_t cap_bytes, size_t *consumed) {
  size_t pos = 0, nout = 0;
  while (npackets - pos >= 12 && cap_bytes - nout >= 204) {
    const uint8_t *pin = in_packets + pos * 204;
    int delay = 0;
    for (int i = 0; i < 204; ++i, delay = (delay + 1) % 12) out[nout + i] = pin[(11 - delay) * 204 + i];
    pos += 1;
    nout += 204;
  }
  *consumed = pos;
  return nout;
}

/* dvb_convol (dvb.h:567-604) on convol_multipoly<uint16_t,16>::encode (convolutional.h:237-264) */
struct lo_convol { int bits_in, bits_out, bps; uint16_t polys[8]; uint16_t hist; int nhist; uint16_t sersymb; int nsersymb; };
lo_convol *lo_convol_new(int rate, int bits_per_symbol) {
  static const uint16_t G1 = 0171, G2 = 0133;
  lo_convol *c = (lo_convol *)calloc(1, sizeof(*c));
  const uint16_t p12[] = {G1, G2}, p23[] = {G1, G2, G2 << 1}, p46[] = {G1, G2, G2 << 1, G1 << 2, G2 << 2, G2 << 3},
                 p34[] = {G1, G2, G2 << 1, G1 << 2}, p45[] = {G1, G2, G2 << 1, G1 << 2, G1 << 3},
                 p56[] = {G1, G2, G2 << 1, G1 << 2, G2 << 3, G1 << 4},
                 p78[] = {G1, G2, G2 << 1, G2 << 2, G2 << 3, G1 << 4, G2 << 5, G1 << 6};
  const uint16_t *p = NULL;
  switch (rate) {   /* fec_specs, dvb.h:556-566 */
    case LO_FEC12: c->bits_in = 1; c->bits_out = 2; p = p12; break;
    case LO_FEC23: c->bits_in = 2; c->bits_out = 3; p = p23; break;
    case LO_FEC46: c->bits_in = 4; c->bits_out = 6; p = p46; break;
    case LO_FEC34: c->bits_in = 3; c->bits_out = 4; p = p34; break;
    case LO_FEC56: c->bits_in = 5; c->bits_out = 6; p = p56; break;
    case LO_FEC78: c->bits_in = 7; c->bits_out = 8; p = p78; break;
    case LO_FEC45: c->bits_in = 4; c->bits_out = 5; p = p45; break;
    default: free(c); return NULL;
  }
  memcpy(c->polys, p, sizeof(uint16_t) * c->bits_out);
  c->bps = bits_per_symbol;
  if (c->bits_out % c->bps) { free(c); return NULL; }   /* "Code rate not suitable for this constellation" */
  return c;
}
void lo_convol_free(lo_convol *c) { free(c); }
/* one run() call, dvb.h:586-597 */
size_t lo_convol_run(lo_convol *c, const uint8_t *in, size_t n_in, uint8_t *out, size_t cap, size_t *consumed) {
  long count = (long)n_in;
  long lim = (long)(cap * c->bps / c->bits_out * c->bits_in / 8);
  if (lim < count) count = lim;
  count = (count / c->bits_in) * c->bits_in;
  const uint8_t symbmask = (uint8_t)((1 << c->bps) - 1);
  uint8_t *pout = out;
  for (long k = 0; k < count; ++k) {
    const uint8_t b = in[k];
    for (int bit = 8; bit--;) {
      c->hist = (uint16_t)((c->hist >> 1) | ((uint16_t)((b >> bit) & 1) << 15));
      ++c->nhist;
      if (c->nhist == c->bits_in) {
        for (int p = 0; p < c->bits_out; ++p) c->sersymb = (uint16_t)((c->sersymb << 1) | (__builtin_parity((uint16_t)(c->hist & c->polys[p])) & 1));
        c->nhist = 0;
        c->nsersymb += c->bits_out;
        while (c->nsersymb >= c->bps) {
          *pout++ = (uint8_t)((c->sersymb >> (c->nsersymb - c->bps)) & symbmask);
          c->nsersymb -= c->bps;
        }
      }
    }
  }
  *consumed = (size_t)count;
  return (size_t)(pout - out);
}

/* cstln_transmitter<f32,0>::run, sdr.h:1206-1218 */
void lo_cstln_transmitter(const lo_cstln_lut *c, const uint8_t *sym, size_t n, lo_cf32 *out) {
  for (size_t k = 0; k < n; ++k) { out[k].re = 0 + c->symbols[sym[k]][0]; out[k].im = 0 + c->symbols[sym[k]][1]; }
}

/* simple_agc<f32>::run, sdr.h:253-273: chunks of 128 */
size_t lo_simple_agc(float *estimated, float out_rms, float bw, const lo_cf32 *in, size_t n, lo_cf32 *out) {
  size_t pos = 0;
  while (n - pos >= 128) {
    float amp2 = 0;
    for (int i = 0; i < 128; ++i) amp2 += in[pos + i].re * in[pos + i].re + in[pos + i].im * in[pos + i].im;
    amp2 /= 128;
    if (!*estimated) *estimated = amp2;
    *estimated = *estimated * (1 - bw) + amp2 * bw;
    const float gain = *estimated ? out_rms / sqrtf(*estimated) : 0;
    for (int i = 0; i < 128; ++i) { out[pos + i].re = in[pos + i].re * gain; out[pos + i].im = in[pos + i].im * gain; }
    pos += 128;
  }
  return pos;
}
