// leansdr_amd/host/apps/leandvb_amd.cc — the leandvb receive graph on MI355X.
//
// Mirrors the front half of the reference's graph builder (src/apps/leandvb.cc:157-515):
//   stdin → [cconverter | scaler] → [fir_filter + decimation (--resample)] | [decimator]
//         → cstln_receiver(sampler) → soft symbols
// with the same option names and the same parameter arithmetic, built on the host
// framework in ../leansdr (reference class surface) whose blocks call the HIP kernels
// through the C ABI.  Device pipebufs are sized by --buf-factor (default 4096: large
// batches, few launches).  Until the FEC tail (deconvolution/Viterbi → mpeg_sync →
// deinterleaver → RS → derandomizer) is wired to its GPU blocks the program writes the
// soft-symbol stream (4-byte softsymbol records) to stdout; `--fd-info N` prints the
// FREQ/SS/MER lines of the reference (leandvb.cc:600-616).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "leansdr/framework.h"
#include "leansdr/generic.h"
#include "leansdr/dsp.h"
#include "leansdr/sdr.h"
#include "leansdr/filtergen.h"
#include "leansdr/dvb.h"

using namespace leansdr;

struct config {
  bool verbose, debug;
  enum { INPUT_U8, INPUT_F32 } input_format;
  float float_scale;
  float Fs, Fm;
  cstln_lut<256>::predef constellation;
  int fec;  // LSDR_FEC*
  float Ftune;
  bool allow_drift;
  bool viterbi;
  int anf;            // auto_notch slots (leandvb default 1)
  bool cnr;
  float Fderot;
  int fd_const;
  bool json;
  bool fastlock;
  bool highspeed;
  int fd_spectrum;
  bool resample;
  float resample_rej;
  int decim;
  enum { SAMP_NEAREST, SAMP_LINEAR, SAMP_RRC } sampler;
  int rrc_steps;
  float rrc_rej, rolloff;
  unsigned long buf_factor;
  int fd_info;
  float Finfo;
  bool out_symbols;   // write the soft-symbol stream instead of TS packets
  bool tiled;
  unsigned tile_len, tile_warmup;
  int device;
  config()
      : verbose(false), debug(false), input_format(INPUT_U8), float_scale(1.0), Fs(2.4e6), Fm(2e6),
        constellation(cstln_lut<256>::QPSK), fec(LSDR_FEC12), Ftune(0), allow_drift(false), viterbi(false),
        anf(1), cnr(false), Fderot(0), fd_const(-1), json(false), fastlock(false), highspeed(false), fd_spectrum(-1), resample(false), resample_rej(10), decim(1), sampler(SAMP_LINEAR), rrc_steps(0), rrc_rej(10), rolloff(0.35),
        buf_factor(4096), fd_info(-1), Finfo(5), out_symbols(false), tiled(false), tile_len(0), tile_warmup(0), device(0) {}
};

// The constants leandvb prints first on --fd-info (leandvb.cc:143-155).
static void output_initial_info(int fd, const config &cfg) {
  static const char *const cstln_name[] = {"BPSK", "QPSK", "8PSK", "16APSK", "32APSK", "64APSKe", "16QAM", "64QAM", "256QAM"};
  static const int rate_in[] = {1, 2, 4, 3, 5, 7, 4, 8, 9}, rate_out[] = {2, 3, 6, 4, 6, 8, 5, 9, 10};   // code_rate order, dvb.h:38-39
  const char *q = cfg.json ? "\"" : "";
  char text[256];
  int n = snprintf(text, sizeof(text), "STANDARD %sDVB-S%s\nCONSTELLATION %s%s%s\nCR %s%d/%d%s\nSR %f\n", q, q, q,
                   cstln_name[cfg.constellation], q, q, rate_in[cfg.fec], rate_out[cfg.fec], q, cfg.Fm);
  if (n > 0 && write(fd, text, n) != n) fatal("write(fd_info)");
}
// VBER window: about twice per second, and fine enough to resolve 2e-5 (leandvb.cc:585-587)
static int vber_window(const config &cfg) {
  int w = cfg.Fm / 2;
  return w < 50000 ? 50000 : w;
}

static int decimation(float Fin, float Fout) {
  int d = Fin / Fout;
  return max(d, 1);
}

// leandvb --hs (leandvb.cc:727-969): fast_qpsk_receiver<u8> → dvb_deconvol_sync_hard → mpeg_sync(fastlock) → … → TS.
static int run_highspeed(config &cfg) {
  scheduler sch;
  sch.verbose = cfg.verbose;
  sch.debug = cfg.debug;
  lsdr_ctx *ctx = NULL;
  lsdr_check(lsdr_ctx_create(cfg.device, NULL, &ctx), "lsdr_ctx_create");
  unsigned long BUF_BASEBAND = 4096 * cfg.buf_factor, BUF_SYMBOLS = 1024 * cfg.buf_factor, BUF_BYTES = 2048 * cfg.buf_factor;
  unsigned long BUF_MPEGBYTES = 2448 * cfg.buf_factor, BUF_PACKETS = cfg.buf_factor, BUF_SLOW = cfg.buf_factor;
  if (cfg.input_format != config::INPUT_U8) fail("--hs requires --u8");
  if (cfg.fec != LSDR_FEC12) fail("--hs currently supports code rate 1/2 only");

  pipebuf<cu8> p_stdin(&sch, "stdin", BUF_BASEBAND);
  file_reader<cu8> r_stdin(&sch, 0, p_stdin);
  pipebuf<cu8> p_rawiq(&sch, "rawiq", BUF_BASEBAND, ctx);
  h2d_copier<cu8> r_h2d(&sch, ctx, p_stdin, p_rawiq);

  pipebuf<f32> p_freq(&sch, "freq", BUF_SLOW);
  pipebuf<u8> p_symbols(&sch, "PSK hard symbols", BUF_SYMBOLS, ctx);
  fast_qpsk_receiver<u8> demod(&sch, p_rawiq, p_symbols, &p_freq, NULL);
  demod.set_omega(cfg.Fs / cfg.Fm);
  if (cfg.Ftune) demod.set_freq(cfg.Ftune / cfg.Fs);
  if (cfg.allow_drift) demod.allow_drift = true;
  demod.meas_decimation = decimation(cfg.Fs, cfg.Finfo);
  demod.tiled = cfg.tiled; demod.tile_len = cfg.tile_len; demod.tile_warmup = cfg.tile_warmup;

  pipebuf<u8> p_bytes(&sch, "bytes", BUF_BYTES, ctx);
  dvb_deconvol_sync_hard r_deconv(&sch, p_symbols, p_bytes);
  r_deconv.resync_period = cfg.fastlock ? 1 : 32;

  pipebuf<u8> p_mpegbytes(&sch, "mpegbytes", BUF_MPEGBYTES, ctx);
  pipebuf<int> p_lock(&sch, "lock", BUF_SLOW);
  pipebuf<u32> p_locktime(&sch, "locktime", BUF_PACKETS);
  mpeg_sync<u8, 0> r_sync(&sch, p_bytes, p_mpegbytes, NULL, &p_lock, &p_locktime);
  r_sync.fastlock = true;
  r_sync.resync_period = cfg.fastlock ? 1 : 32;

  pipebuf<rspacket<u8> > p_rspackets(&sch, "RS-enc packets", BUF_PACKETS, ctx);
  deinterleaver<u8> r_deinter(&sch, p_mpegbytes, p_rspackets);
  pipebuf<int> p_vbitcount(&sch, "Bits processed", BUF_PACKETS);
  pipebuf<int> p_verrcount(&sch, "Bits corrected", BUF_PACKETS);
  pipebuf<tspacket> p_rtspackets(&sch, "rand TS packets", BUF_PACKETS, ctx);
  rs_decoder<u8, 0> r_rsdec(&sch, p_rspackets, p_rtspackets, &p_vbitcount, &p_verrcount);
  pipebuf<float> p_vber(&sch, "VBER", BUF_SLOW);
  rate_estimator<float> r_vber(&sch, p_verrcount, p_vbitcount, p_vber);
  r_vber.sample_size = vber_window(cfg);
  pipebuf<tspacket> p_tspackets(&sch, "TS packets", BUF_PACKETS, ctx);
  derandomizer r_derand(&sch, p_rtspackets, p_tspackets);
  pipebuf<tspacket> p_ts_host(&sch, "TS packets(host)", BUF_PACKETS);
  d2h_copier<tspacket> r_d2h(&sch, ctx, p_tspackets, p_ts_host);
  file_writer<tspacket> r_stdout(&sch, p_ts_host, 1);

  if (cfg.fd_info >= 0) {   // leandvb.cc:897-910
    (new file_printer<f32>(&sch, "FREQ %.0f\n", p_freq, cfg.fd_info))->scale = cfg.Fs;
    new file_printer<int>(&sch, "LOCK %d\n", p_lock, cfg.fd_info);
    new file_printer<u32>(&sch, "LOCKTIME %lu\n", p_locktime, cfg.fd_info, decimation(cfg.Fm / 8 / 204, cfg.Finfo));
    new file_printer<float>(&sch, "VBER %.6f\n", p_vber, cfg.fd_info);
    output_initial_info(cfg.fd_info, cfg);
  }
  sch.run();
  sch.shutdown();
  lsdr_ctx_destroy(ctx);
  return 0;
}

static int run(config &cfg) {
  scheduler sch;
  sch.verbose = cfg.verbose;
  sch.debug = cfg.debug;
  lsdr_ctx *ctx = NULL;
  lsdr_check(lsdr_ctx_create(cfg.device, NULL, &ctx), "lsdr_ctx_create");

  // Buffer sizes follow leandvb.cc:185-202.
  unsigned long BUF_BASEBAND = 4096 * cfg.buf_factor;
  unsigned long BUF_SYMBOLS = 1024 * cfg.buf_factor;
  unsigned long BUF_SLOW = cfg.buf_factor;

  // INPUT (host pipebuf) → HBM
  pipebuf<cf32> *p_preprocessed = NULL;
  float fuse_scale = 0;
  pipebuf<cu8> *p_rawu8 = NULL;
  if (cfg.input_format == config::INPUT_U8) {
    pipebuf<cu8> *p_stdin = new pipebuf<cu8>(&sch, "stdin", BUF_BASEBAND);
    new file_reader<cu8>(&sch, 0, *p_stdin);
    p_rawu8 = new pipebuf<cu8>(&sch, "stdin(hbm)", BUF_BASEBAND, ctx);
    new h2d_copier<cu8>(&sch, ctx, *p_stdin, *p_rawu8);
    pipebuf<cf32> *p_rawiq = new pipebuf<cf32>(&sch, "rawiq", BUF_BASEBAND, ctx);
    new cconverter<u8, 128, f32, 0, 1, 1>(&sch, *p_rawu8, *p_rawiq);
    p_preprocessed = p_rawiq;
  } else {
    pipebuf<cf32> *p_stdin = new pipebuf<cf32>(&sch, "stdin", BUF_BASEBAND);
    new file_reader<cf32>(&sch, 0, *p_stdin);
    pipebuf<cf32> *p_dev = new pipebuf<cf32>(&sch, "stdin(hbm)", BUF_BASEBAND, ctx);
    new h2d_copier<cf32>(&sch, ctx, *p_stdin, *p_dev);
    if (cfg.resample) {
      fuse_scale = cfg.float_scale;  // scaler fused into the fir_filter's load stage
      p_preprocessed = p_dev;
    } else {
      pipebuf<cf32> *p_rawiq = new pipebuf<cf32>(&sch, "rawiq", BUF_BASEBAND, ctx);
      new scaler<float, cf32, cf32>(&sch, cfg.float_scale, *p_dev, *p_rawiq);
      p_preprocessed = p_rawiq;
    }
  }

  // NOTCH FILTER (leandvb.cc:294-306).  A fused scaler cannot sit behind the notch: materialise it first.
  if (cfg.anf) {
    if (fuse_scale) {
      pipebuf<cf32> *p_rawiq = new pipebuf<cf32>(&sch, "rawiq", BUF_BASEBAND, ctx);
      new scaler<float, cf32, cf32>(&sch, cfg.float_scale, *p_preprocessed, *p_rawiq);
      p_preprocessed = p_rawiq;
      fuse_scale = 0;
    }
    pipebuf<cf32> *p_autonotched = new pipebuf<cf32>(&sch, "autonotched", BUF_BASEBAND, ctx);
    auto_notch<f32> *r_anf = new auto_notch<f32>(&sch, *p_preprocessed, *p_autonotched, cfg.anf, 0);
    if (cfg.tiled && cfg.anf <= 4) r_anf->set_throughput_mode();
    p_preprocessed = p_autonotched;
  } else if (cfg.verbose) fprintf(stderr, "ANF is disabled (requires a clean signal).\n");

  // FREQUENCY CORRECTION (leandvb.cc:308-318)
  if (cfg.Fderot) {
    if (fuse_scale) {
      pipebuf<cf32> *p_rawiq = new pipebuf<cf32>(&sch, "rawiq", BUF_BASEBAND, ctx);
      new scaler<float, cf32, cf32>(&sch, cfg.float_scale, *p_preprocessed, *p_rawiq);
      p_preprocessed = p_rawiq;
      fuse_scale = 0;
    }
    if (cfg.verbose) fprintf(stderr, "Derotating from %.3f kHz\n", cfg.Fderot / 1e3);
    pipebuf<cf32> *p_derot = new pipebuf<cf32>(&sch, "derotated", BUF_BASEBAND, ctx);
    new rotator<f32>(&sch, *p_preprocessed, *p_derot, -cfg.Fderot / cfg.Fs);
    p_preprocessed = p_derot;
  }

  // CNR ESTIMATION (leandvb.cc:320-329)
  pipebuf<f32> p_cnr(&sch, "cnr", BUF_SLOW);
  cnr_fft<f32> *r_cnr = NULL;
  if (cfg.cnr) {
    r_cnr = new cnr_fft<f32>(&sch, *p_preprocessed, p_cnr, cfg.Fm / cfg.Fs);
    r_cnr->decimation = decimation(cfg.Fs, 1);  // 1 Hz
  }

  // SPECTRUM (leandvb.cc:331-343)
  pipebuf<f32[1024]> *p_spectrum = NULL;
  if (cfg.fd_spectrum >= 0) {
    p_spectrum = new pipebuf<f32[1024]>(&sch, "spectrum", BUF_SLOW);
    spectrum<f32> *r_spectrum = new spectrum<f32>(&sch, *p_preprocessed, *p_spectrum);
    r_spectrum->decimation = decimation(cfg.Fs, 1);  // 1 Hz
    r_spectrum->kavg = 0.5;
  }

  // FILTERING (leandvb.cc:353-384)
  fir_filter<cf32, float> *r_resample = NULL;
  int decim = 1;
  if (cfg.resample) {
    if (cfg.decim > 1) decim = cfg.decim;
    else {
      float target_Fs = cfg.Fm * 4;
      decim = cfg.Fs / target_Fs;
      if (decim < 1) decim = 1;
    }
    float transition = (cfg.Fm / 2) * cfg.rolloff;
    int order = cfg.resample_rej * cfg.Fs / (22 * transition);
    order = ((order + 1) / 2) * 2;
    if (cfg.verbose) fprintf(stderr, "Inserting filter: order %d, decimation %d.\n", order, decim);
    pipebuf<cf32> *p_resampled = new pipebuf<cf32>(&sch, "resampled", BUF_BASEBAND, ctx);
    float *coeffs;
    float Fcut = (cfg.Fm / 2) * (1 + cfg.rolloff / 2) / cfg.Fs;
    int ncoeffs = filtergen::lowpass(order, Fcut, &coeffs);
    filtergen::normalize_dcgain(ncoeffs, coeffs, 1);
    r_resample = new fir_filter<cf32, float>(&sch, ncoeffs, coeffs, *p_preprocessed, *p_resampled, decim, fuse_scale);
    p_preprocessed = p_resampled;
    cfg.Fs /= decim;
  } else if (cfg.decim > 1) {
    decim = cfg.decim;
    pipebuf<cf32> *p_decimated = new pipebuf<cf32>(&sch, "decimated", BUF_BASEBAND, ctx);
    new decimator<cf32>(&sch, decim, *p_preprocessed, *p_decimated);
    p_preprocessed = p_decimated;
    cfg.Fs /= decim;
  }

  // RECEIVER (leandvb.cc:425-502)
  pipebuf<softsymbol> p_symbols(&sch, "PSK soft-symbols", BUF_SYMBOLS, ctx);
  pipebuf<f32> p_freq(&sch, "freq", BUF_SLOW);
  pipebuf<f32> p_ss(&sch, "SS", BUF_SLOW);
  pipebuf<f32> p_mer(&sch, "MER", BUF_SLOW);
  sampler_interface<f32> *sampler;
  switch (cfg.sampler) {
    case config::SAMP_NEAREST: sampler = new nearest_sampler<float>(); break;
    case config::SAMP_LINEAR: sampler = new linear_sampler<float>(); break;
    default: {
      float *coeffs;
      if (cfg.rrc_steps == 0) cfg.rrc_steps = max(1, (int)(64 * cfg.Fm / cfg.Fs));
      float Frrc = cfg.Fs * cfg.rrc_steps;
      float transition = (cfg.Fm / 2) * cfg.rolloff;
      int order = cfg.rrc_rej * Frrc / (22 * transition);
      int ncoeffs = filtergen::root_raised_cosine(order, cfg.Fm / Frrc, cfg.rolloff, &coeffs);
      if (cfg.verbose) fprintf(stderr, "RRC interpolator: %d steps, %d coeffs.\n", cfg.rrc_steps, ncoeffs);
      sampler = new fir_sampler<float, float>(ncoeffs, coeffs, cfg.rrc_steps);
    }
  }
  pipebuf<cf32> p_sampled(&sch, "PSK symbols", BUF_BASEBAND);   // host pipe: one constellation point per chunk (leandvb.cc:451)
  cstln_receiver<f32> demod(&sch, sampler, *p_preprocessed, p_symbols, &p_freq, &p_ss, &p_mer,
                            cfg.fd_const >= 0 ? &p_sampled : NULL);
  demod.cstln = new cstln_lut<256>(cfg.constellation, cfg.fec);
  demod.set_omega(cfg.Fs / cfg.Fm);
  if (cfg.Ftune) demod.set_freq(cfg.Ftune / cfg.Fs);
  if (cfg.allow_drift) demod.set_allow_drift(true);
  if (cfg.viterbi) demod.pll_adjustment /= 6;
  demod.meas_decimation = decimation(cfg.Fs, cfg.Finfo);
  if (cfg.tiled) {
    demod.mode = LSDR_RX_TILED;
    demod.tile_len = cfg.tile_len;
    demod.tile_warmup = cfg.tile_warmup;
  }

  if (r_cnr) { r_cnr->freq_tap = &demod.freq_tap; r_cnr->tap_multiplier = 1.0 / decim; }   // leandvb.cc:512-515

  // TRACKING FILTERS (leandvb.cc:506-510): the receiver→filter feedback edge stays on the host.
  if (r_resample) {
    r_resample->freq_tap = &demod.freq_tap;
    r_resample->tap_multiplier = 1.0 / decim;
    r_resample->freq_tol = cfg.Fm / (cfg.Fs * decim) * 0.1;
  }

  unsigned long BUF_BYTES = 2048 * cfg.buf_factor;
  unsigned long BUF_MPEGBYTES = 2448 * cfg.buf_factor;
  unsigned long BUF_PACKETS = cfg.buf_factor;
  pipebuf<int> p_lock(&sch, "lock", BUF_SLOW);
  pipebuf<u32> p_locktime(&sch, "locktime", BUF_PACKETS);
  pipebuf<int> p_vbitcount(&sch, "Bits processed", BUF_PACKETS);
  pipebuf<int> p_verrcount(&sch, "Bits corrected", BUF_PACKETS);
  pipebuf<float> p_vber(&sch, "VBER", BUF_SLOW);
  if (cfg.out_symbols) {
    // soft symbols back to the host, to stdout
    pipebuf<softsymbol> *p_symbols_host = new pipebuf<softsymbol>(&sch, "soft-symbols(host)", BUF_SYMBOLS);
    new d2h_copier<softsymbol>(&sch, ctx, p_symbols, *p_symbols_host);
    new file_writer<softsymbol>(&sch, *p_symbols_host, 1);
  } else {
    // DECONVOLUTION AND SYNCHRONIZATION … TS OUTPUT (leandvb.cc:519-596), all in HBM
    pipebuf<u8> *p_bytes = new pipebuf<u8>(&sch, "bytes", BUF_BYTES, ctx);
    deconvol_sync_simple *r_deconv = NULL;
    if (cfg.viterbi) {
      if (cfg.fec == FEC23 && (demod.cstln->nsymbols == 4 || demod.cstln->nsymbols == 64)) cfg.fec = FEC46;   // leandvb.cc:533-537
      viterbi_sync *r = new viterbi_sync(&sch, p_symbols, *p_bytes, demod.cstln, (code_rate)cfg.fec);
      if (cfg.fastlock) r->resync_period = 1;   // leandvb.cc:540
    } else {
      r_deconv = make_deconvol_sync_simple(&sch, p_symbols, *p_bytes, (code_rate)cfg.fec);
      r_deconv->fastlock = cfg.fastlock;        // leandvb.cc:543
    }
    pipebuf<u8> *p_mpegbytes = new pipebuf<u8>(&sch, "mpegbytes", BUF_MPEGBYTES, ctx);
    mpeg_sync<u8, 0> *r_sync = new mpeg_sync<u8, 0>(&sch, *p_bytes, *p_mpegbytes, r_deconv, &p_lock, &p_locktime);
    r_sync->fastlock = cfg.fastlock;            // leandvb.cc:565
    pipebuf<rspacket<u8> > *p_rspackets = new pipebuf<rspacket<u8> >(&sch, "RS-enc packets", BUF_PACKETS, ctx);
    new deinterleaver<u8>(&sch, *p_mpegbytes, *p_rspackets);
    pipebuf<tspacket> *p_rtspackets = new pipebuf<tspacket>(&sch, "rand TS packets", BUF_PACKETS, ctx);
    new rs_decoder<u8, 0>(&sch, *p_rspackets, *p_rtspackets, &p_vbitcount, &p_verrcount);
    (new rate_estimator<float>(&sch, p_verrcount, p_vbitcount, p_vber))->sample_size = vber_window(cfg);
    pipebuf<tspacket> *p_tspackets = new pipebuf<tspacket>(&sch, "TS packets", BUF_PACKETS, ctx);
    new derandomizer(&sch, *p_rtspackets, *p_tspackets);
    pipebuf<tspacket> *p_ts_host = new pipebuf<tspacket>(&sch, "TS packets(host)", BUF_PACKETS);
    new d2h_copier<tspacket>(&sch, ctx, *p_tspackets, *p_ts_host);
    new file_writer<tspacket>(&sch, *p_ts_host, 1);
  }

  if (cfg.fd_info >= 0) {   // same printers, same order (= order of the lines within a scheduler pass) as leandvb.cc:600-616
    (new file_printer<f32>(&sch, "FREQ %.0f\n", p_freq, cfg.fd_info))->scale = cfg.Fs;
    new file_printer<f32>(&sch, "SS %f\n", p_ss, cfg.fd_info);
    new file_printer<f32>(&sch, "MER %.1f\n", p_mer, cfg.fd_info);
    new file_printer<int>(&sch, "LOCK %d\n", p_lock, cfg.fd_info);
    new file_printer<u32>(&sch, "LOCKTIME %lu\n", p_locktime, cfg.fd_info, decimation(cfg.Fm / 8 / 204, cfg.Finfo));
    new file_printer<f32>(&sch, "CNR %.1f\n", p_cnr, cfg.fd_info);
    new file_printer<f32>(&sch, "VBER %.6f\n", p_vber, cfg.fd_info);
    output_initial_info(cfg.fd_info, cfg);
  } else {
    // unread measurement pipes never block their writer (pipebuf with zero readers packs to empty)
  }

  if (cfg.fd_const >= 0) {   // leandvb.cc:617-645
    cstln_lut<256> *c = demod.cstln;
    FILE *f = fdopen(dup(cfg.fd_const), "w");
    if (!f) fatal("fdopen(fd_const)");
    if (cfg.json) {
      fprintf(f, "CONST [");
      for (int i = 0; i < c->nsymbols; ++i) fprintf(f, "%s[%d,%d]", i ? "," : "", c->symbols[i].re, c->symbols[i].im);
      fprintf(f, "]\n");
    } else {
      fprintf(f, "CONST %d", c->nsymbols);
      for (int i = 0; i < c->nsymbols; ++i) fprintf(f, " %d,%d", c->symbols[i].re, c->symbols[i].im);
      fprintf(f, "\n");
    }
    fclose(f);
    file_carrayprinter<f32> *symbol_printer =
        cfg.json ? new file_carrayprinter<f32>(&sch, "SYMBOLS [", "[%.0f,%.0f]", ",", "]\n", p_sampled, cfg.fd_const)
                 : new file_carrayprinter<f32>(&sch, "SYMBOLS %d", " %.0f,%.0f", "", "\n", p_sampled, cfg.fd_const);
    symbol_printer->fixed_size = 128;
  }
  if (cfg.fd_spectrum >= 0)   // leandvb.cc:647-652
    new file_vectorprinter<f32, 1024>(&sch, "SPECTRUM [", "%.3f", ",", "]\n", *p_spectrum, cfg.fd_spectrum);

  sch.run();
  sch.shutdown();
  if (cfg.debug) sch.dump();
  lsdr_ctx_destroy(ctx);
  return 0;
}

static void usage(const char *name, FILE *f, int c) {
  fprintf(f,
          "Usage: %s [options]  < IQ  > TS\n"
          "MI355X build of the leandvb DVB-S receive path.\n"
          "  --u8 | --f32           input format (default u8)\n"
          "  --float-scale FLOAT    scale for --f32 input\n"
          "  -f HZ, --sr HZ         sample rate, symbol rate\n"
          "  --const STRING         QPSK (default), BPSK, 8PSK, 16APSK, 32APSK\n"
          "  --cr STRING            1/2 (default), 2/3, 3/4, 5/6, 7/8 (APSK radii)\n"
          "  --anf N, --cnr         auto-notch slots (default 1, 0 disables), CNR estimator\n"
          "  --fastlock, --hq       synchronise more aggressively; --hq = --fastlock --viterbi --sampler rrc\n"
          "  --hs                   high-speed path: --u8 QPSK 1/2, all-integer receiver (leandvb --hs)\n"
          "  --derotate HZ          frequency-shift the preprocessed signal (rotator)\n"
          "  --fd-const FD [--json] CONST / SYMBOLS lines (constellation and sampled points)\n"
          "  --fd-spectrum FD       SPECTRUM [..1024 dB values..] lines, one per second of signal\n"
          "  --tune HZ, --drift     receiver bias, unlimited drift\n"
          "  --resample, --resample-rej FLOAT, --decim N, --roll-off FLOAT\n"
          "  --sampler nearest|linear|rrc, --rrc-steps N, --rrc-rej FLOAT\n"
          "  --viterbi              PLL parameters for low SNR (pll_adjustment/6)\n"
          "  --buf-factor N         pipebuf scale (default 4096)\n"
          "  --tiled [--tile-len N --tile-warmup N]   throughput mode of the receiver\n"
          "  --out-symbols          write soft symbols instead of TS packets\n"
          "  --fd-info FD, --device N, -v, -d\n",
          name);
  exit(c);
}

int main(int argc, const char *argv[]) {
  config cfg;
  for (int i = 1; i < argc; ++i) {
    const char *a = argv[i];
    auto need = [&](void) -> const char * { if (i + 1 >= argc) usage(argv[0], stderr, 1); return argv[++i]; };
    if (!strcmp(a, "-h")) usage(argv[0], stdout, 0);
    else if (!strcmp(a, "-v")) cfg.verbose = true;
    else if (!strcmp(a, "-d")) cfg.debug = true;
    else if (!strcmp(a, "--u8")) cfg.input_format = config::INPUT_U8;
    else if (!strcmp(a, "--f32")) cfg.input_format = config::INPUT_F32;
    else if (!strcmp(a, "--float-scale")) cfg.float_scale = atof(need());
    else if (!strcmp(a, "-f")) cfg.Fs = atof(need());
    else if (!strcmp(a, "--sr")) cfg.Fm = atof(need());
    else if (!strcmp(a, "--tune")) cfg.Ftune = atof(need());
    else if (!strcmp(a, "--drift")) cfg.allow_drift = true;
    else if (!strcmp(a, "--viterbi")) cfg.viterbi = true;
    else if (!strcmp(a, "--resample")) cfg.resample = true;
    else if (!strcmp(a, "--resample-rej")) cfg.resample_rej = atof(need());
    else if (!strcmp(a, "--decim")) cfg.decim = atoi(need());
    else if (!strcmp(a, "--roll-off")) cfg.rolloff = atof(need());
    else if (!strcmp(a, "--rrc-steps")) cfg.rrc_steps = atoi(need());
    else if (!strcmp(a, "--rrc-rej")) cfg.rrc_rej = atof(need());
    else if (!strcmp(a, "--buf-factor")) cfg.buf_factor = atol(need());
    else if (!strcmp(a, "--fd-info")) cfg.fd_info = atoi(need());
    else if (!strcmp(a, "--device")) cfg.device = atoi(need());
    else if (!strcmp(a, "--anf")) cfg.anf = atoi(need());
    else if (!strcmp(a, "--cnr")) cfg.cnr = true;
    else if (!strcmp(a, "--derotate")) cfg.Fderot = atof(need());
    else if (!strcmp(a, "--fd-const")) cfg.fd_const = atoi(need());
    else if (!strcmp(a, "--json")) cfg.json = true;
    else if (!strcmp(a, "--fastlock")) cfg.fastlock = true;
    else if (!strcmp(a, "--hs")) cfg.highspeed = true;
    else if (!strcmp(a, "--hq")) { cfg.fastlock = true; cfg.viterbi = true; cfg.sampler = config::SAMP_RRC; }   // leandvb.cc:1154-1158
    else if (!strcmp(a, "--fd-spectrum")) cfg.fd_spectrum = atoi(need());
    else if (!strcmp(a, "--tiled")) cfg.tiled = true;
    else if (!strcmp(a, "--out-symbols")) cfg.out_symbols = true;
    else if (!strcmp(a, "--tile-len")) cfg.tile_len = atoi(need());
    else if (!strcmp(a, "--tile-warmup")) cfg.tile_warmup = atoi(need());
    else if (!strcmp(a, "--sampler")) {
      const char *s = need();
      if (!strcmp(s, "nearest")) cfg.sampler = config::SAMP_NEAREST;
      else if (!strcmp(s, "linear")) cfg.sampler = config::SAMP_LINEAR;
      else if (!strcmp(s, "rrc")) cfg.sampler = config::SAMP_RRC;
      else usage(argv[0], stderr, 1);
    } else if (!strcmp(a, "--const")) {
      const char *s = need();
      static const struct { const char *n; cstln_lut<256>::predef v; } tab[] = {
          {"BPSK", cstln_lut<256>::BPSK}, {"QPSK", cstln_lut<256>::QPSK}, {"8PSK", cstln_lut<256>::PSK8},
          {"16APSK", cstln_lut<256>::APSK16}, {"32APSK", cstln_lut<256>::APSK32}, {"64APSKe", cstln_lut<256>::APSK64E},
          {"16QAM", cstln_lut<256>::QAM16}, {"64QAM", cstln_lut<256>::QAM64}, {"256QAM", cstln_lut<256>::QAM256}};
      bool ok = false;
      for (auto &t : tab) if (!strcmp(s, t.n)) { cfg.constellation = t.v; ok = true; }
      if (!ok) usage(argv[0], stderr, 1);
    } else if (!strcmp(a, "--cr")) {
      const char *s = need();
      static const struct { const char *n; int v; } tab[] = {{"1/2", LSDR_FEC12}, {"2/3", LSDR_FEC23}, {"4/6", LSDR_FEC46},
          {"3/4", LSDR_FEC34}, {"5/6", LSDR_FEC56}, {"7/8", LSDR_FEC78}, {"4/5", LSDR_FEC45}, {"8/9", LSDR_FEC89}, {"9/10", LSDR_FEC910}};
      bool ok = false;
      for (auto &t : tab) if (!strcmp(s, t.n)) { cfg.fec = t.v; ok = true; }
      if (!ok) usage(argv[0], stderr, 1);
    } else usage(argv[0], stderr, 1);
  }
  return cfg.highspeed ? run_highspeed(cfg) : run(cfg);
}
