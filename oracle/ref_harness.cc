// oracle/ref_harness.cc — TEST INFRASTRUCTURE, not product code.
//
// Our own driver TU around the *real* reference (pabr/leansdr).  It #includes
// the reference headers from where they lie (-I/root/reference/src; nothing is
// copied into this repository) and exposes every hot-path block of SURVEY.md
// §8(a) behind a plain C function, so that tests/ and oracle/make_golden.py
// can (1) validate the plain-C restatement in oracle/lsdr_oracle*.c and
// (2) generate the golden vectors committed under tests/golden/.
//
// Built only where /root/reference exists (oracle/Makefile) into
// oracle/_ref/libleansdr_ref.so.  Single TU on purpose: the reference headers
// define non-inline functions (framework.h:32-33, math.h:56-91).
//
// Each function wires   buffer_reader -> <reference block> -> buffer_writer
// (generic.h:336-375) on a private scheduler and runs it to its fixpoint
// (framework.h:96-104), exactly the way leandvb.cc:715 does.

#define private public      // expose lookup tables for fixture dumps
#define protected public
#include "leansdr/framework.h"
#include "leansdr/generic.h"
#include "leansdr/dsp.h"
#include "leansdr/sdr.h"
#include "leansdr/dvb.h"
#include "leansdr/rs.h"
#include "leansdr/filtergen.h"
#undef private
#undef protected

using namespace leansdr;

namespace {
// Pipebuf sizes follow leandvb.cc:185-202 with the default buf_factor=4.
const unsigned long BUF_BASEBAND = 4096 * 4;
const unsigned long BUF_SYMBOLS = 1024 * 4;
const unsigned long BUF_BYTES = 2048 * 4;
const unsigned long BUF_MPEGBYTES = 2448 * 4;
const unsigned long BUF_PACKETS = 4;
const unsigned long BUF_SLOW = 4;
}  // namespace

extern "C" {

// ---------------------------------------------------------------- tables

// math.h:95-111
void ref_trig16(float *out /* [65536][2] */) {
  trig16 *t = new trig16();
  memcpy(out, t->lut, sizeof(t->lut));
  delete t;
}

// index semantics of trig16::expi(float), math.h:108-110
unsigned ref_trig16_index(float a) {
  static trig16 *t = new trig16();
  return (unsigned)(&t->expi(a) - t->lut);
}

// sdr.h:313-573.  out: 65536 records {int16 cost; uint8 symbol; uint8 pad(0); int16 phase_error}
// written as three separate arrays to avoid padding ambiguity.
int ref_cstln_lut(int predef, float g1, float g2, float g3, int harden,
                  int16_t *cost, uint8_t *symbol, int16_t *phase_error,
                  int8_t *symbols /* [256][2] */, int *nrotations) {
  cstln_lut<256> *c =
      new cstln_lut<256>((cstln_lut<256>::predef)predef, g1, g2, g3);
  if (harden) c->harden();
  for (int i = 0; i < 256; ++i)
    for (int q = 0; q < 256; ++q) {
      cost[i * 256 + q] = c->lut[i][q].ss.cost;
      symbol[i * 256 + q] = c->lut[i][q].ss.symbol;
      phase_error[i * 256 + q] = c->lut[i][q].phase_error;
    }
  for (int s = 0; s < c->nsymbols; ++s) {
    symbols[2 * s] = c->symbols[s].re;
    symbols[2 * s + 1] = c->symbols[s].im;
  }
  *nrotations = c->nrotations;
  int n = c->nsymbols;
  return n;
}

// float lookup incl. the halving loop, sdr.h:470-482
void ref_cstln_lookup(int predef, float I, float Q, int16_t *cost,
                      uint8_t *symbol, int16_t *pe) {
  static cstln_lut<256> *c[16];
  if (!c[predef]) c[predef] = new cstln_lut<256>((cstln_lut<256>::predef)predef);
  cstln_lut<256>::result *r = c[predef]->lookup(I, Q);
  *cost = r->ss.cost;
  *symbol = r->ss.symbol;
  *pe = r->phase_error;
}

// dvb.h:45-81 (radii per code rate for APSK; plain for QPSK/8PSK)
int ref_make_dvbs2_constellation(int predef, int fec, int16_t *cost,
                                 uint8_t *symbol, int16_t *phase_error,
                                 int8_t *symbols) {
  cstln_lut<256> *c = make_dvbs2_constellation((cstln_lut<256>::predef)predef,
                                               (code_rate)fec);
  for (int i = 0; i < 256; ++i)
    for (int q = 0; q < 256; ++q) {
      cost[i * 256 + q] = c->lut[i][q].ss.cost;
      symbol[i * 256 + q] = c->lut[i][q].ss.symbol;
      phase_error[i * 256 + q] = c->lut[i][q].phase_error;
    }
  for (int s = 0; s < c->nsymbols; ++s) {
    symbols[2 * s] = c->symbols[s].re;
    symbols[2 * s + 1] = c->symbols[s].im;
  }
  return c->nsymbols;
}

// filtergen.h:45-62 followed by the extra normalize_dcgain of leandvb.cc:377-378
int ref_lowpass(int order, float Fcut, int renormalize, float *out) {
  float *c;
  int n = filtergen::lowpass(order, Fcut, &c);
  if (renormalize) filtergen::normalize_dcgain(n, c, 1);
  memcpy(out, c, n * sizeof(float));
  delete[] c;
  return n;
}

// filtergen.h:68-92
int ref_root_raised_cosine(int order, float Fs, float rolloff, float *out) {
  float *c;
  int n = filtergen::root_raised_cosine(order, Fs, rolloff, &c);
  memcpy(out, c, n * sizeof(float));
  delete[] c;
  return n;
}

// filtergen.h:26-32
void ref_normalize_power(int n, float *c, float gain) {
  filtergen::normalize_power(n, c, gain);
}

// ---------------------------------------------------------------- streaming blocks

// dsp.h:33-54   cconverter<u8,128,f32,0,1,1>
long ref_cconverter_u8(const uint8_t *in, long n, float *out) {
  scheduler sch;
  pipebuf<cu8> p_in(&sch, "in", BUF_BASEBAND);
  pipebuf<cf32> p_out(&sch, "out", BUF_BASEBAND);
  buffer_reader<cu8> r(&sch, (cu8 *)in, n, p_in);
  cconverter<u8, 128, f32, 0, 1, 1> c(&sch, p_in, p_out);
  buffer_writer<cf32> w(&sch, p_out, (cf32 *)out, n);
  sch.run();
  return w.pos;
}

// dsp.h:140-160   scaler<float,cf32,cf32>
long ref_scaler(float scale, const float *in, long n, float *out) {
  scheduler sch;
  pipebuf<cf32> p_in(&sch, "in", BUF_BASEBAND);
  pipebuf<cf32> p_out(&sch, "out", BUF_BASEBAND);
  buffer_reader<cf32> r(&sch, (cf32 *)in, n, p_in);
  scaler<float, cf32, cf32> c(&sch, scale, p_in, p_out);
  buffer_writer<cf32> w(&sch, p_out, (cf32 *)out, n);
  sch.run();
  return w.pos;
}

// dsp.h:219-285   fir_filter<cf32,float>.  A non-zero `freq` is applied the way
// leandvb does it: through the freq_tap pointer, which makes the first run()
// call set_freq(freq) (dsp.h:236-244) before any output is produced.
long ref_fir_filter(int ncoeffs, const float *coeffs, unsigned decim, float freq,
                    const float *in, long n, float *out, long cap,
                    float *shifted_out /* [ncoeffs][2] or NULL */) {
  scheduler sch;
  pipebuf<cf32> p_in(&sch, "in", BUF_BASEBAND);
  pipebuf<cf32> p_out(&sch, "out", BUF_BASEBAND);
  buffer_reader<cf32> r(&sch, (cf32 *)in, n, p_in);
  fir_filter<cf32, float> f(&sch, ncoeffs, (float *)coeffs, p_in, p_out, decim);
  float tap = freq;
  if (freq != 0) {
    f.freq_tap = &tap;
    f.tap_multiplier = 1;
    f.freq_tol = 0;
  }
  buffer_writer<cf32> w(&sch, p_out, (cf32 *)out, cap);
  sch.run();
  if (shifted_out) memcpy(shifted_out, f.shifted_coeffs, ncoeffs * sizeof(cf32));
  return w.pos;
}

// dsp.h:290-364   fir_resampler<cf32,float> (interpolator)
long ref_fir_resampler(int ncoeffs, const float *coeffs, int interp, float freq,
                       const float *in, long n, float *out, long cap) {
  scheduler sch;
  pipebuf<cf32> p_in(&sch, "in", BUF_BASEBAND);
  pipebuf<cf32> p_out(&sch, "out", BUF_BASEBAND * 4 + interp);
  buffer_reader<cf32> r(&sch, (cf32 *)in, n, p_in);
  fir_resampler<cf32, float> f(&sch, ncoeffs, (float *)coeffs, p_in, p_out, interp, 1);
  float tap = freq;
  if (freq != 0) {
    f.freq_tap = &tap;
    f.tap_multiplier = 1;
    f.freq_tol = 0;
  }
  buffer_writer<cf32> w(&sch, p_out, (cf32 *)out, cap);
  sch.run();
  return w.pos;
}

// generic.h:247-267
long ref_decimator(int d, const float *in, long n, float *out, long cap) {
  scheduler sch;
  pipebuf<cf32> p_in(&sch, "in", BUF_BASEBAND);
  pipebuf<cf32> p_out(&sch, "out", BUF_BASEBAND);
  buffer_reader<cf32> r(&sch, (cf32 *)in, n, p_in);
  decimator<cf32> f(&sch, d, p_in, p_out);
  buffer_writer<cf32> w(&sch, p_out, (cf32 *)out, cap);
  sch.run();
  return w.pos;
}

// sdr.h:46-154   auto_notch<f32>; `decimation` is the public tunable (sdr.h:54)
long ref_auto_notch(int nslots, int decimation, float k, float agc_rms_setpoint,
                    const float *in, long n, float *out, int *slot_bins) {
  scheduler sch;
  pipebuf<cf32> p_in(&sch, "in", BUF_BASEBAND);
  pipebuf<cf32> p_out(&sch, "out", BUF_BASEBAND);
  buffer_reader<cf32> r(&sch, (cf32 *)in, n, p_in);
  auto_notch<f32> a(&sch, p_in, p_out, nslots, agc_rms_setpoint);
  a.decimation = decimation;
  a.k = k;
  // The reference never initialises slot.estim / slot.expj before the first
  // detect() (sdr.h:57-62,145-150).  In leandvb the block is built on a fresh
  // heap, i.e. zeros (bit-exact pass-through until the first
  // detect); inside this long-lived harness the heap is recycled, so make the
  // fresh-heap condition explicit.
  for (int s = 0; s < nslots; ++s) {
    a.slots[s].estim = cf32(0, 0);
    memset(a.slots[s].expj, 0, sizeof(cf32) * a.fft.n);
  }
  buffer_writer<cf32> w(&sch, p_out, (cf32 *)out, n);
  sch.run();
  if (slot_bins)
    for (int s = 0; s < nslots; ++s) slot_bins[s] = a.slots[s].i;
  return w.pos;
}

// dsp.h:56-116
void ref_cfft(int n, float *data, int reverse) {
  cfft_engine<float> fft(n);
  fft.inplace((cf32 *)data, reverse != 0);
}

// sdr.h:1273-1345   cnr_fft<f32>
long ref_cnr_fft(float bandwidth, int nfft, int decimation, float freq_tap,
                 float tap_multiplier, const float *in, long n, float *out, long cap) {
  scheduler sch;
  pipebuf<cf32> p_in(&sch, "in", BUF_BASEBAND);
  pipebuf<float> p_out(&sch, "out", 1024);
  buffer_reader<cf32> r(&sch, (cf32 *)in, n, p_in);
  cnr_fft<f32> c(&sch, p_in, p_out, bandwidth, nfft);
  c.decimation = decimation;
  float tap = freq_tap;
  c.freq_tap = &tap;
  c.tap_multiplier = tap_multiplier;
  buffer_writer<float> w(&sch, p_out, out, cap);
  sch.run();
  return w.pos;
}

// ---------------------------------------------------------------- cstln_receiver

struct ref_rx_params {
  int sampler;           // 0 nearest, 1 linear, 2 fir (rrc)
  int ncoeffs;           // fir sampler
  const float *coeffs;   // fir sampler prototype
  int subsampling;       // fir sampler
  int cstln;             // cstln_lut<256>::predef
  int fec;               // code_rate for make_dvbs2_constellation
  float omega;           // samples per symbol
  float freq;            // initial set_freq()  (cycles/sample)
  float pll_adjustment;  // 1, or 1/6 with --viterbi (leandvb.cc:498-501)
  int allow_drift;
  unsigned long meas_decimation;
  float kest;
};

struct ref_rx_state {    // final private state, for state-parity checks
  float mu, phase, freqw, agc_gain, est_insp, est_sp, est_ep, freq_tap;
  float min_freqw, max_freqw;
  unsigned long meas_count;
  float hist[12];
};

// sdr.h:697-938.  Outputs: soft symbols as separate cost/symbol arrays (the pad
// byte of `softsymbol` is indeterminate), plus the measurement pipes.
long ref_cstln_receiver(const ref_rx_params *p, const float *in, long n,
                        int16_t *cost, uint8_t *symbol, long cap,
                        float *freq_out, float *ss_out, float *mer_out,
                        float *cstln_out, long meas_cap, long *n_meas, long *n_cstln,
                        ref_rx_state *st) {
  scheduler sch;
  pipebuf<cf32> p_in(&sch, "in", BUF_BASEBAND);
  pipebuf<softsymbol> p_sym(&sch, "sym", BUF_SYMBOLS);
  pipebuf<f32> p_freq(&sch, "freq", BUF_SLOW + 64);
  pipebuf<f32> p_ss(&sch, "ss", BUF_SLOW + 64);
  pipebuf<f32> p_mer(&sch, "mer", BUF_SLOW + 64);
  pipebuf<cf32> p_sampled(&sch, "sampled", BUF_BASEBAND);
  buffer_reader<cf32> r(&sch, (cf32 *)in, n, p_in);
  sampler_interface<f32> *sampler;
  switch (p->sampler) {
    case 0: sampler = new nearest_sampler<float>(); break;
    case 1: sampler = new linear_sampler<float>(); break;
    default:
      sampler = new fir_sampler<float, float>(p->ncoeffs, (float *)p->coeffs,
                                              p->subsampling);
  }
  cstln_receiver<f32> demod(&sch, sampler, p_in, p_sym, &p_freq, &p_ss, &p_mer,
                            &p_sampled);
  demod.cstln = make_dvbs2_constellation((cstln_lut<256>::predef)p->cstln,
                                         (code_rate)p->fec);
  demod.set_omega(p->omega);
  if (p->freq) demod.set_freq(p->freq);
  demod.set_allow_drift(p->allow_drift != 0);
  demod.pll_adjustment = p->pll_adjustment;
  demod.meas_decimation = p->meas_decimation;
  demod.kest = p->kest;
  softsymbol *tmp = new softsymbol[cap];
  buffer_writer<softsymbol> w(&sch, p_sym, tmp, cap);
  buffer_writer<f32> wf(&sch, p_freq, freq_out, meas_cap);
  buffer_writer<f32> ws(&sch, p_ss, ss_out, meas_cap);
  buffer_writer<f32> wm(&sch, p_mer, mer_out, meas_cap);
  buffer_writer<cf32> wc(&sch, p_sampled, (cf32 *)cstln_out, *n_cstln);
  sch.run();
  for (long i = 0; i < w.pos; ++i) {
    cost[i] = tmp[i].cost;
    symbol[i] = tmp[i].symbol;
  }
  delete[] tmp;
  *n_meas = wf.pos;
  *n_cstln = wc.pos;
  if (st) {
    st->mu = demod.mu; st->phase = demod.phase; st->freqw = demod.freqw;
    st->agc_gain = demod.agc_gain; st->est_insp = demod.est_insp;
    st->est_sp = demod.est_sp; st->est_ep = demod.est_ep;
    st->freq_tap = demod.freq_tap;
    st->min_freqw = demod.min_freqw; st->max_freqw = demod.max_freqw;
    st->meas_count = demod.meas_count;
    memcpy(st->hist, demod.hist, sizeof(st->hist));
  }
  return w.pos;
}

}  // extern "C"
